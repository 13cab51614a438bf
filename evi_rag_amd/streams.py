"""HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask).

Why: the hot kernels of this path are sized to own a CU — a scan workgroup holds all of a CU's vector registers, a scorer GEMM
workgroup 128 KiB of its LDS — so a stage made of many SMALL kernels (the PyTorch encoder forward for a batch of questions)
cannot slip in beside them from another stream: it queues behind whole workgroups.  Giving that stage a few CUs of its own, and
keeping the big kernels' streams off them, lets it run BESIDE them.  MI355X has 256 CUs in 8 XCDs; 16 CUs cost the
bandwidth-bound scan about their share of the chip and hide a 2.3 ms stage.

The masks are per stream and per process; nothing about the device is changed.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

_HIP = None
_LIVE: list = []  # (ctypes stream handle) kept so that the handles outlive the torch wrappers


def _hip():
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
        _HIP.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
        _HIP.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        _HIP.hipStreamDestroy.restype = ctypes.c_int
        _HIP.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    return _HIP


def cu_count(device: torch.device) -> int:
    return int(torch.cuda.get_device_properties(device).multi_processor_count)


def masked_stream(device: torch.device, cus: Sequence[int]) -> torch.cuda.Stream:
    """A stream whose kernels run only on the CUs listed (indices in [0, cu_count))."""
    n = cu_count(device)
    cus = sorted({int(c) for c in cus})
    if not cus or cus[0] < 0 or cus[-1] >= n:
        raise ValueError(f"CU indices must be a non-empty subset of [0, {n}), got {cus[:4]}...")
    words = (n + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in cus:
        mask[c // 32] |= 1 << (c % 32)
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = _hip().hipExtStreamCreateWithCUMask(ctypes.byref(handle), words, mask)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with code {rc}")
    _LIVE.append(handle)
    return torch.cuda.ExternalStream(handle.value, device=device)


def partitioned_streams(device: torch.device, shares: Sequence[Optional[int]]) -> List[torch.cuda.Stream]:
    """One stream per entry of `shares`: an int = that many CUs of its own (taken from CU 0 upwards), None = all the CUs no int
    entry took (the None entries share them).  [16, None, None]: 16 CUs for the first stream, the other 240 for the two others."""
    n = cu_count(device)
    taken = 0
    ranges = []
    for s in shares:
        if s is None:
            ranges.append(None)
            continue
        s = int(s)
        if s <= 0 or taken + s >= n:
            raise ValueError(f"cannot reserve {s} CUs (already {taken} of {n} taken; the shared pool needs at least one)")
        ranges.append(range(taken, taken + s))
        taken += s
    rest = range(taken, n)
    return [masked_stream(device, rest if r is None else r) for r in ranges]


def release() -> None:
    """Destroy the masked streams created so far (call when no work is pending on them)."""
    while _LIVE:
        _hip().hipStreamDestroy(_LIVE.pop())


__all__ = ["masked_stream", "partitioned_streams", "cu_count", "release"]
