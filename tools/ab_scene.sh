#!/bin/bash
# A/B builds of libptrace.so with tools/scene_rate.py: tools/ab_scene.sh "<scene_rate args>" lib1.so ...  (paths relative to csrc/)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
ARGS=$1; shift
for lib in base "$@"; do
  if [ "$lib" = base ]; then unset PTRACE_LIB; else export PTRACE_LIB=$lib; fi      # loaded through PTRACE_LIB (_ffi.py): the installed library is not touched
  for rep in 1 2; do echo -n "$lib: "; timeout -k 5 90 python -u tools/scene_rate.py $ARGS || exit 1; done
done
