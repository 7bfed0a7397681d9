"""Descriptor tables for egr_repack_f32 (include/egorear_train.h): every layout change between nn.Parameters and the
kernels' operand layouts of one training step in a single launch."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np
import torch

from .hip import _launch, _pv, _stream, lib

COPYPAD, FWD, DGRAD, UNPACK = 0, 1, 2, 3
_BLOCK = 1024

_DT = np.dtype([("src", "<u8"), ("dst", "<u8"), ("kind", "<i4"), ("rows", "<i4"), ("cin", "<i4"), ("cin_tot", "<i4"), ("ci0", "<i4"),
                ("cin_pad", "<i4"), ("taps", "<i4"), ("rows_pad", "<i4"), ("k_off", "<i4"), ("k_tot", "<i4"), ("total", "<i8")])
assert _DT.itemsize == 64

lib.egr_repack_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
lib.egr_repack_f32.restype = C.c_int


class RepackTable:
    """Host-side list of descriptors -> device table + block map.  Tensors named by descriptors are kept alive here."""

    def __init__(self, device):
        self.dev = device
        self.rows: List[tuple] = []
        self.keep: List[torch.Tensor] = []
        self.table = None
        self.blocks = None
        self.n_blocks = 0

    def add(self, kind: int, src: torch.Tensor, dst: torch.Tensor, dst_off: int, *, rows: int, cin: int = 0, cin_tot: int = 0, ci0: int = 0,
            cin_pad: int = 0, taps: int = 1, rows_pad: int = 0, k_off: int = 0, k_tot: int = 0, total: int = 0, src_off: int = 0):
        """src / dst: contiguous fp32 tensors on the device; *_off in elements."""
        if src.dtype != torch.float32 or dst.dtype != torch.float32 or not src.is_cuda or not dst.is_cuda:
            raise RuntimeError("egorear_amd.repack: fp32 device tensors expected")
        if total <= 0:
            return
        self.rows.append((src.data_ptr() + 4 * src_off, dst.data_ptr() + 4 * dst_off, kind, rows, cin, cin_tot, ci0, cin_pad, taps, rows_pad,
                          k_off, k_tot, total))
        self.keep += [src, dst]
        self.table = None

    def __len__(self):
        return len(self.rows)

    def finalize(self):
        arr = np.array(self.rows, dtype=_DT)
        nb = [(int(r[-1]) + _BLOCK - 1) // _BLOCK for r in self.rows]
        blocks = np.empty((sum(nb), 2), dtype=np.int64)
        k = 0
        for i, n in enumerate(nb):
            blocks[k:k + n, 0] = i
            blocks[k:k + n, 1] = np.arange(n, dtype=np.int64) * _BLOCK
            k += n
        # pinned staging + asynchronous upload: a pageable copy would drain the stream (the host would wait for the whole
        # reverse pass before it could enqueue the optimiser); the pinned tensors stay alive with the table
        self._host = (torch.from_numpy(arr.view(np.uint8).copy()).pin_memory(), torch.from_numpy(blocks).pin_memory())
        self.table = self._host[0].to(self.dev, non_blocking=True)
        self.blocks = self._host[1].to(self.dev, non_blocking=True)
        self.n_blocks = int(blocks.shape[0])

    def run(self):
        if not self.rows:
            return
        if self.table is None:
            self.finalize()
        _launch("egr_repack_f32", lib.egr_repack_f32, _pv(self.table), _pv(self.blocks), self.n_blocks, _stream())
