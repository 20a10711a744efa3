#!/bin/bash
# tools/r05_txfinal.sh: the transmit half's evidence in one run (tests, 8-seed soak, both hosts' benches, the phase probe, rocprofv3)
mkdir -p gpurun_out/r05
export TMPDIR=/tmp; R=$(pwd)
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_udp.py tests/test_gpu_txchain.py -x -q -m gpu > gpurun_out/r05/gputests_tx_final.log 2>&1; rc=$?; tail -3 gpurun_out/r05/gputests_tx_final.log; [ $rc = 0 ] || { grep -n "^E" gpurun_out/r05/gputests_tx_final.log | head; exit 1; }
bash tools/r05_tx_soak.sh > gpurun_out/r05/tx_soak.txt 2>&1; cat gpurun_out/r05/tx_soak.txt
for a in "128 8" "128 8" "128 1" "128 2" "128 4" "512 8" "512 8" "128 8 copy"; do timeout -k 10 120 python tools/group_tx_bench.py $a || exit 1; done > gpurun_out/r05/group_tx_bench_final.txt 2>&1
make -C openbts-ttsou_amd/csrc tx_bench > /dev/null 2>&1
for a in "128 8" "128 8" "128 8" "128 1" "128 4" "512 8" "512 8" "1024 8"; do timeout -k 10 120 openbts-ttsou_amd/tx_bench $a || exit 1; done > gpurun_out/r05/group_tx_bench_cpp_final.txt 2>&1
TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_txprobe.so timeout -k 10 200 python tools/group_tx_probe.py 128 8 > gpurun_out/r05/tx_probe_final.txt 2>&1
TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_txprobe.so timeout -k 10 200 python tools/group_tx_probe.py 128 1 >> gpurun_out/r05/tx_probe_final.txt 2>&1
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/prof_txf -- $R/openbts-ttsou_amd/tx_bench 128 8 > $R/gpurun_out/r05/group_tx_bench_cpp_under_rocprof.json 2> $R/gpurun_out/r05/prof_txf.log || exit 1
python3 $R/tools/prof_summary.py $R/gpurun_out/r05/prof_txf > $R/gpurun_out/r05/kernel_stats_tx_final.csv
rm -rf $R/gpurun_out/r05/prof_txf
cat $R/gpurun_out/r05/kernel_stats_tx_final.csv | cut -c1-160
grep -v amdgpu $R/gpurun_out/r05/group_tx_bench_cpp_final.txt | cut -c50-330
grep -v amdgpu $R/gpurun_out/r05/tx_probe_final.txt | cut -c1-700
