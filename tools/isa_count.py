#!/usr/bin/env python3
"""Development helper: per-kernel instruction mix of the largest basic block (the steady-state loop) of a hipcc -save-temps .s file.
    python tools/isa_count.py gpurun_out/save_temps/lean_core-hip-amdgcn-amd-amdhsa-gfx950.s [name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
names = [n for n in re.findall(r'^(_Z\w+):\s*(?:;.*)?$', s, re.M) if want in n]
KEYS = ['ds_read_b32', 'ds_read_b64', 'ds_read_b128', 'v_mfma_f32_4x4x4_16b_f16', 'v_mfma_f32_16x16x32_f16', 'v_mfma_f32_32x32x16_f16', 'v_mov_b32_dpp',
        'v_perm_b32', 'v_lshlrev_b32_sdwa', 'v_lshlrev_b32_e32', 'v_and_b32_e32', 'v_bfe_u32', 'v_lshl_or_b32', 'v_exp_f32_e32', 'global_load_dwordx4',
        'v_mov_b32_e32', 's_waitcnt', 's_nop', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32', 'scratch_load_dword', 'scratch_store_dword']
for name in names:
    tail = s[s.index(name + ':'):]
    end = re.search(r'^\.Lfunc_end\d+:', tail, re.M)
    if not end:
        continue
    body = tail[:end.start()]
    m = re.search(r'; NumVgprs: (\d+)', tail)
    sc = re.search(r'; ScratchSize: (\d+)', tail)
    blocks = re.split(r'^\.LBB\d+_\d+:.*$', body, flags=re.M)
    big = max(blocks, key=len)
    ops = collections.Counter(l.split()[0] for l in big.splitlines() if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':'))
    valu = sum(v for k, v in ops.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    print(f"{name}: VGPRs {m.group(1) if m else '?'} scratch {sc.group(1) if sc else '?'}; largest block {sum(ops.values())} instructions ({valu} vector ALU), {len(blocks)} blocks")
    print("   ", {k: ops[k] for k in KEYS if ops[k]})
    print("   ", sorted(((k, v) for k, v in ops.items() if k not in KEYS), key=lambda x: -x[1])[:24])
