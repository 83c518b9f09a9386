#!/bin/bash
# A/B of libptrace builds on the instancing demo (GPU box): tools/ab_inst.sh lib1.so lib2.so ... (names in csrc/; loaded
# through PTRACE_LIB)
for lib in "$@"; do
  echo "== $lib"
  PTRACE_LIB=$lib timeout -k 10 300 python -u tools/inst_demo.py $DEMO_ARGS || exit 1
done
