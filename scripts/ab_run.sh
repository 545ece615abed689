#!/bin/bash
# On the GPU box: time a command under each libmrx_hip.so variant of ab/.
#   scripts/ab_run.sh "<command>" name1 name2 ...
set -eu
CMD="$1"; shift
LIB=madrona_renderer_amd/libmrx_hip.so
cp $LIB /tmp/libmrx_hip.so.keep
for n in "$@"; do
  cp "ab/libmrx_hip.so.$n" $LIB
  echo "== $n"
  timeout -k 10 200 bash -c "$CMD"
done
cp /tmp/libmrx_hip.so.keep $LIB
