#!/bin/bash
# the GPU test-suite (+ optional extra pytest args).  gpurun_out/$1/pytest.log
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-suite}; shift
mkdir -p $out
python -m pytest tests -m gpu -q --timeout 1500 "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -40 $out/pytest.log
