#!/bin/bash
mkdir -p gpurun_out/r05
P=openbts-ttsou_amd/csrc/build_probe
timeout -k 10 600 python -m pytest tests/test_gpu_normal.py tests/test_gpu_soft_tolerance.py tests/test_gpu_trxgroup.py -x -q > gpurun_out/r05/gputests_h.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/gputests_h.log; tail -4 gpurun_out/r05/gputests_h.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
{
for r in 1 2 3; do
echo "# wide"; timeout -k 10 200 python tools/tol_sweep.py --steps 300 --skip-tuning || exit 1
echo "# narrow"; TRXSIG_LIB=$P/libtrxsig_cw0.so timeout -k 10 200 python tools/tol_sweep.py --steps 300 --skip-tuning || exit 1
done
} > gpurun_out/r05/tol_sweep4.txt 2> gpurun_out/r05/tol_sweep4.err
rc2=$?; grep -v beside gpurun_out/r05/tol_sweep4.txt | cut -c1-200; tail -3 gpurun_out/r05/tol_sweep4.err
exit $rc2
