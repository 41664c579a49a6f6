#!/bin/bash
# usage: scratch/pmc_run.sh <tag> <B> -- counters...   (one rocprofv3 --pmc pass of the microbench; db lands in gpurun_out/pmc_<tag>)
tag=$1; B=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/pmc_$tag -o run -- python scratch/microbench.py $B 50 > gpurun_out/pmc_$tag.log 2>&1
ls gpurun_out/pmc_$tag | head -3
