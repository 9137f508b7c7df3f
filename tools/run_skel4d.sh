#!/bin/bash
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export SKEL4_SWEEP5=1
timeout 900 build/skel4 16 12 > $out/skeleton4_sweep5.txt 2>&1
cat $out/skeleton4_sweep5.txt
