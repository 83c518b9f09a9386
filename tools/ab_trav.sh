#!/bin/bash
# Vote sweep of the traversal-phase pool builds on the GPU box: tools/ab_trav.sh lib.so ...
export VOTES="56,56,32,48 56,56,64,48 56,56,16,48 56,56,32,32 56,56,32,56 40,56,32,48 64,48,32,48"
exec bash tools/ab_pool.sh "$@"
