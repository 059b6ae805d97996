# development: time the prompt-attention kernel of every A/B library under build/ab/ given on the command line
for m in "$@"; do echo "EXP=$m"; MILLION_HIP_LIB=build/ab/libmillion_exp$m.so timeout -k 10 120 python tools/prefill_once_time.py 2>&1 | tail -1; done
