"""Device picker of the search loop on ROCm.

The reference's ``GPUManager`` (src_1gp/utils.py:185-246, used by glam.py to place one run per free GPU) parses
``nvidia-smi --query-gpu`` output, which does not exist on an MI355X host.  Same surface — ``GPUManager(qargs)``,
``.gpus`` (dicts with ``index``, ``gpu_name``, ``memory.free``, ``memory.total`` in MiB), ``.auto_choice(thre)``,
``.wait_free_gpu(thre)`` — answered by the HIP runtime (``torch.cuda.mem_get_info``), no subprocess, no arithmetic on
the hot path.
"""
from __future__ import annotations

import time

import torch

_MIB = 1 << 20


class GPUManager:
    def __init__(self, qargs=()):
        self.qargs = list(qargs)
        self.gpus = self.query_gpu(self.qargs)
        self.gpu_num = len(self.gpus)

    @staticmethod
    def query_gpu(qargs=()):
        """One dict per visible device, keys as in the reference's query (extra ``qargs`` it cannot answer are ``None``)."""
        out = []
        for i in range(torch.cuda.device_count() if torch.cuda.is_available() else 0):
            free, total = torch.cuda.mem_get_info(i)
            info = {"index": str(i), "gpu_name": torch.cuda.get_device_name(i), "memory.free": free // _MIB,
                    "memory.total": total // _MIB}
            info.update({k: None for k in qargs if k not in info})
            out.append(info)
        return out

    @staticmethod
    def _sort_by_memory(gpus, by_size=False):
        if by_size:
            return sorted(gpus, key=lambda d: d["memory.free"], reverse=True)
        return sorted(gpus, key=lambda d: float(d["memory.free"]) / d["memory.total"], reverse=True)

    def auto_choice(self, thre):
        """Index of the device with the most free memory, or ``None`` when its free fraction is below ``thre``."""
        for old, new in zip(self.gpus, self.query_gpu(self.qargs)):
            old.update(new)
        if not self.gpus:
            return None
        best = self._sort_by_memory(self.gpus, True)[0]
        if float(best["memory.free"]) / best["memory.total"] < thre:
            return None
        return int(best["index"])

    def wait_free_gpu(self, thre=0.7, poll_seconds=30):
        if not torch.cuda.is_available():
            return -1                          # the reference's CPU answer (utils.py:216-217)
        while True:
            choice = self.auto_choice(thre)
            if choice is not None:
                return choice
            print("Keep Looking @ {}".format(time.asctime(time.localtime(time.time()))))
            time.sleep(poll_seconds)
