"""One-process-per-GPU sharded tick (BASELINE config 4): local scan -> all-gather of the per-shard top-k lists ->
merge + accept decision on every rank.  Plumbing only: the scan and the merge are HIP kernels behind
chip_scan_local / chip_merge_decide; the exchange is torch.distributed (backend "nccl" == RCCL over xGMI on the GPU
box, "gloo" in the CPU tests, where a stand-in backend object plays the device).

Row -> rank map (must match chip_internal.h): global row i lives on rank i % G at local index i // G, so every prefix
[0, k) is balanced to within one row.  Exchange payload per tick and rank: 3 queries x K x (f64 score, i64 index) =
384 B at K = 8 -- latency-bound, one small collective per tick.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import capi

ENTRY_BYTES = 16  # chip_topk_entry {double score; int64 idx}


def owner_of(row: int, world: int) -> int:
    return row % world


def local_index(row: int, world: int) -> int:
    return row // world


def local_count(k: int, rank: int, world: int) -> int:
    """number of rows of the global prefix [0, k) stored on `rank`"""
    return (k - rank + world - 1) // world if k > rank else 0


def global_index(local: int, rank: int, world: int) -> int:
    return local * world + rank


class ShardedLoopDetector:
    """`device_api` is a capi.Chip created with shard_rank/shard_count (or any object with the same scan_local /
    merge_decide / set_stream methods)."""

    def __init__(self, device_api, topk: int = capi.CHIP_DEFAULT_TOPK, group=None, device: str | torch.device = "cuda"):
        self.api = device_api
        self.topk = topk
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        # raw bytes of chip_topk_entry[3][K]; float64 is only a convenient 8-byte carrier (idx are int64 bit patterns).
        # local-merge -> all-gather -> global-merge of a tick are all ordered on ONE stream, so one buffer set suffices;
        # the device API overlaps them with the NEXT tick's scan (which runs on the ctx's internal scan stream).
        self.local = torch.zeros((3, topk, 2), dtype=torch.float64, device=self.device)
        self.gathered = torch.zeros((self.world * 3, topk, 2), dtype=torch.float64, device=self.device)
        self.stream = None
        if self.device.type == "cuda":
            # A dedicated torch stream is made "current" around every collective: torch.distributed orders the RCCL
            # kernel after / before that stream, and the ctx enqueues its merges and stream waits on the same one.
            # The scans themselves run on the ctx's internal scan stream (overlap with the previous tick's exchange).
            self.stream = torch.cuda.Stream(self.device)
            self.api.set_stream(self.stream.cuda_stream)

    def _scan_and_gather(self, l: int, params):
        status = self.api.scan_local(l, self.local.data_ptr(), self.topk, params)
        if status == capi.CHIP_TICK_SCANNED:
            # concatenation form (world*3, K, 2): accepted by both the RCCL and the gloo backends
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
            else:
                dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
        return status

    @staticmethod
    def _not_scanned(status):
        r = capi.TickResult()
        r.status = status
        r.idx_curr = r.idx_prev = -1
        return r

    def tick(self, l: int, params=None):
        """One pass of Cerebro::descrip_N__dot__descrip_0_N's loop body at l.  Every rank returns the same record."""
        status = self._scan_and_gather(l, params)
        if status != capi.CHIP_TICK_SCANNED:
            return self._not_scanned(status)
        return self.api.merge_decide(l, self.gathered.data_ptr(), self.world, self.topk, params)

    # ---- pipelined form: no host synchronisation between ticks; collect(slot) later, in enqueue order ----
    def tick_enqueue(self, l: int, slot: int, params=None):
        status = self._scan_and_gather(l, params)
        if status == capi.CHIP_TICK_SCANNED:
            self.api.merge_decide_enqueue(l, self.gathered.data_ptr(), self.world, slot, self.topk, params)
        return status

    def collect(self, slot: int):
        return self.api.loop_tick_collect(slot)

    def close(self):
        if self.stream is not None:
            self.stream.synchronize()
            self.api.set_stream(None)
            self.stream = None
