#!/bin/bash
# tools/ab_script.sh <other lib in csrc/> <python script> : run a script with the current and with another build of libptrace.so
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
cp pathtracing_amd/csrc/libptrace.so /tmp/libptrace_cur.so
echo "other ($1):"; cp pathtracing_amd/csrc/$1 pathtracing_amd/csrc/libptrace.so; timeout -k 10 200 python -u $2; rc=$?
cp /tmp/libptrace_cur.so pathtracing_amd/csrc/libptrace.so
[ $rc -eq 0 ] || exit $rc
echo "current:"; timeout -k 10 200 python -u $2
