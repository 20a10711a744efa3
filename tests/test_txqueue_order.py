"""The transmit priority queue's ORDER: the reference queues bursts in a std::priority_queue of pointers
compared by timestamp (CommonLibs/Interthread.h:432-528, Transceiver/radioInterface.h:58-72), so bursts with equal timestamps
leave in an order the heap's shape decides.  Four implementations must agree on every pop of random write / read scripts with
many ties: std::priority_queue itself (tests/txqueue_order.cpp, this image's libstdc++ -- what the one-ARFCN object uses), the
array form of the queue (csrc/trxsig_txq.h), the form the Transceiver group's kernels keep in LDS (csrc/trxsig_txq_lds.h: the same
moves with fewer dependent reads; its array must equal the array form's after every operation) -- both compiled into the same
program for the host -- and the
Python model's heapq (oracle/transceiver_model.py)."""
import heapq
import os
import subprocess

import numpy as np
import pytest

import transceiver_model as tm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("txq") / "txqueue_order")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "openbts-ttsou_amd", "csrc"),
                           os.path.join(ROOT, "tests", "txqueue_order.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("seed,base_fn", [(1, 1000), (2, tm.HYPERFRAME - 40), (3, 0), (4, 77777)])
def test_three_queues_agree(prog, seed, base_fn):
    rng = np.random.default_rng(seed)
    script, ops = [], []
    live = 0
    for i in range(6000):
        if live == 0 or (rng.random() < 0.56 and live < 250):
            fn = int(base_fn + rng.integers(0, 12)) % tm.HYPERFRAME      # few distinct times: ties everywhere, some across the wrap
            tn = int(rng.integers(0, 8))
            script.append("a %d %d %d" % (fn, tn, i)); ops.append(("a", fn, tn, i)); live += 1
        else:
            script.append("p"); ops.append(("p",)); live -= 1
    script += ["p"] * (live + 2); ops += [("p",)] * (live + 2)
    out = subprocess.run([prog], input="\n".join(script) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
    heap, k = [], 0
    for op in ops:
        if op[0] == "a":
            heapq.heappush(heap, tm.Queued((op[1], op[2]), op[3]))
        else:
            mine = heapq.heappop(heap).payload if heap else -1
            std_id, arr_id, lds_id = (int(v) for v in out[k].split()); k += 1
            assert std_id == arr_id == mine and lds_id == (mine & 2047 if mine >= 0 else -1), (k, std_id, arr_id, lds_id, mine)
    assert k == sum(1 for op in ops if op[0] == "p")
