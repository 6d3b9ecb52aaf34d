#!/bin/bash
# One intermediate GPU check, parameterised (what the rounds' numbered check scripts each did by hand):
# a subset of the GPU suite first, then A/B lines (tools/ab.py: one process per library / engine
# configuration, with checksums), then any other commands.  Everything lands in gpurun_out/<tag>/.
#
# usage: tools/gpu_check.sh <tag> [-k '<pytest -k expression>'] [-e 'ENV=VAL ...'] [-a '<ab.py arguments>'] ... [-c '<command>'] ...
#   e.g. gpurun -- tools/gpu_check.sh tail -k 'tail or marginal' \
#            -a '--config C3L --mode marginal - build_variants/libqmhip_x.so' -a '--config C1 --steps 20 -' \
#            -c 'python tools/diag_stream.py C1 8'
cd "$(dirname "$0")/.."
TAG=${1:?tag}; shift
O=gpurun_out/$TAG; mkdir -p $O
ENVS=""
while [ $# -gt 0 ]; do
  case "$1" in
    -k) ( time env $ENVS python -m pytest tests -m gpu -x -q -k "$2" 2>&1 | tail -8 ) 2>&1 | tee -a $O/tests.txt; shift 2 ;;
    -e) ENVS="$2"; shift 2 ;;
    -a) eval "env $ENVS python tools/ab.py $2" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt; shift 2 ;;
    -c) eval "env $ENVS $2" 2>&1 | grep -v amdgpu.ids | tee -a $O/commands.txt; shift 2 ;;
    *) echo "unknown argument $1"; exit 2 ;;
  esac
done
