#!/bin/bash
# round 5, second GPU session: the whole GPU suite (env switches became setters), smoke, the default bench line
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gputests_a.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r05/gputests_a.log
tail -15 gpurun_out/r05/gputests_a.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke_a.log 2>&1; rc1=$?; tail -2 gpurun_out/r05/smoke_a.log
if [ $rc1 -eq 124 ] || [ $rc1 -eq 137 ]; then exit $rc1; fi
timeout -k 10 300 python bench.py --steps 20 --check > gpurun_out/r05/bench_a.json 2> gpurun_out/r05/bench_a.err
rc2=$?
cut -c1-1500 gpurun_out/r05/bench_a.json; tail -3 gpurun_out/r05/bench_a.err
exit $rc2
