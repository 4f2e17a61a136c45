#!/bin/bash
# the GPU test-suite (or the test files / pytest args given).  gpurun_out/$1/pytest.log
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-suite}; shift
mkdir -p $out
target=tests
for a in "$@"; do case "$a" in tests/*) target="";; esac; done
python -m pytest $target -m gpu -q --timeout 1500 "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -40 $out/pytest.log
