#!/bin/bash
# tools/ab_script.sh <other lib in csrc/> <python script> : run a script with another build of libptrace.so (loaded through
# PTRACE_LIB) and with the installed one
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
echo "other ($1):"; PTRACE_LIB=$1 timeout -k 10 200 python -u $2 || exit $?
echo "current:"; timeout -k 10 200 python -u $2
