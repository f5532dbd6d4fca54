"""Peer-visible device buffers for the hand-written collectives (one-shot all-reduce, on-device all-to-all-v).

What a peer GPU reads or writes must not be ordinary caching-allocator memory: that is coarse-grained, and a flag a REMOTE GPU writes
may be served to its spinning owner from the owner's L2 for ever (RCCL and torch's symmetric memory allocate their signal pads uncached
for the same reason; two processes on one GPU share the L2 and cannot show the difference).  `exchange()` therefore allocates through
the C ABI -- `ao_peer_alloc` = hipExtMallocWithFlags, kind 0 uncached (flag blocks), kind 1 fine-grained (staging) -- and swaps raw
`hipIpcMemHandle_t`s over the process group.  If that fails on ANY rank (an HSA build whose IPC refuses such allocations, ...), every
rank falls back together to the round-3 path (torch.zeros + torch's CUDA-IPC storage sharing: coarse-grained) and `mode` says so, so
that callers can keep such a set-up off multi-GPU defaults.
"""
import ctypes
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist

UNCACHED, FINEGRAINED = 0, 1


class _RawView:
    """__cuda_array_interface__ over a raw device pointer (torch.as_tensor builds a non-owning uint8 tensor from it)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class PeerBuffer:
    """One own allocation: `.ptr`, `.nbytes`, `.tensor` (uint8 view for staging copies), freed with the object."""

    def __init__(self, lib, check, nbytes: int, kind: int, device):
        self._lib, self._check = lib, check
        p = ctypes.c_void_p()
        check(lib.ao_peer_alloc(ctypes.byref(p), nbytes, kind))
        self.ptr, self.nbytes, self.kind = int(p.value), int(nbytes), kind
        self.tensor = torch.as_tensor(_RawView(self.ptr, self.nbytes), device=device)

    def handle(self) -> bytes:
        h = ctypes.create_string_buffer(self._lib.ao_peer_handle_bytes())
        self._check(self._lib.ao_peer_export(ctypes.c_void_p(self.ptr), h))
        return h.raw

    def __del__(self):
        try:
            if self.ptr:
                self._lib.ao_peer_free(ctypes.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class _Imported:
    def __init__(self, lib, ptr):
        self._lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            if self.ptr:
                self._lib.ao_peer_close(ctypes.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:  # noqa: BLE001
            pass


def exchange(specs: Sequence[Tuple[int, int]], group, device, force_fallback: bool = False):
    """Allocate one zero-filled buffer per (nbytes, kind) spec on this rank and map every peer's.  Returns
    (own tensors (uint8), ptrs[spec][rank] device pointers as ints, keep-alive objects, mode) with mode "uncached+fine-grained" or
    "coarse-grained fallback: <reason>".  Collective over `group` (two object all-gathers at most)."""
    from . import _lib

    lib, check = _lib.lib(), _lib.check
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    own, handles, why = [], None, None
    if not force_fallback:
        try:
            own = [PeerBuffer(lib, check, n, kind, device) for n, kind in specs]
            handles = [b.handle() for b in own]
        except Exception as e:  # noqa: BLE001
            why, own, handles = f"{type(e).__name__}: {e}", [], None
    else:
        why = "forced"
    gathered = [None] * world
    dist.all_gather_object(gathered, (handles, why), group=group)
    keep: List[object] = []
    if all(g[0] is not None for g in gathered):
        ptrs = [[0] * world for _ in specs]
        try:
            for r in range(world):
                for i, b in enumerate(own):
                    if r == rank:
                        ptrs[i][r] = b.ptr
                    else:
                        p = ctypes.c_void_p()
                        check(lib.ao_peer_import(gathered[r][0][i], ctypes.byref(p)))
                        keep.append(_Imported(lib, int(p.value)))
                        ptrs[i][r] = int(p.value)
            ok_here, err = True, None
        except Exception as e:  # noqa: BLE001
            ok_here, err = False, f"{type(e).__name__}: {e}"
        flags = [None] * world
        dist.all_gather_object(flags, (ok_here, err), group=group)
        if all(f[0] for f in flags):
            keep.extend(own)
            return [b.tensor for b in own], ptrs, keep, "uncached+fine-grained"
        why = next(f[1] for f in flags if not f[0])
        keep.clear()
    else:
        why = next(g[1] for g in gathered if g[0] is None)
    # together: torch's allocator + CUDA-IPC storage sharing (coarse-grained memory; verified with two processes on one GPU only)
    own_t = [torch.zeros(n, dtype=torch.uint8, device=device) for n, _ in specs]
    torch.cuda.synchronize(device)
    mine = tuple(t.untyped_storage()._share_cuda_() for t in own_t)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine, group=group)
    ptrs = [[0] * world for _ in specs]
    for r in range(world):
        for i, t in enumerate(own_t):
            if r == rank:
                ptrs[i][r] = t.data_ptr()
            else:
                st = torch.UntypedStorage._new_shared_cuda(*gathered[r][i])
                keep.append(st)
                ptrs[i][r] = st.data_ptr()
    return own_t, ptrs, keep, f"coarse-grained fallback: {why}"


def _device_ident(host, device, props=None):
    """(host, uuid, pci domain, bus, device) of a GPU, or None when neither the UUID nor the PCI ids say which GPU it is -- a missing,
    empty or all-zero UUID together with missing / negative PCI ids must read as "unknown", never as "the same GPU as everybody else"."""
    try:
        props = props if props is not None else torch.cuda.get_device_properties(device)
    except Exception:  # noqa: BLE001
        return None
    uuid = str(getattr(props, "uuid", "") or "")
    if not any(c not in "0-{} " for c in uuid.replace("GPU", "").replace("gpu", "")):
        uuid = ""
    pci = tuple(int(getattr(props, name, -1)) for name in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    pci_known = pci[1] >= 0 and pci[2] >= 0
    if not uuid and not pci_known:
        return None
    return (host, uuid, pci if pci_known else None)


def spans_devices(group, device) -> bool:
    """True when the ranks of `group` sit on more than one physical GPU (or that cannot be established).  Collective (one object
    all-gather): every rank names its host and the GPU's UUID / PCI bus id."""
    import socket

    ident = _device_ident(socket.gethostname(), device)
    gathered = [None] * dist.get_world_size(group)
    dist.all_gather_object(gathered, ident, group=group)
    return any(g is None for g in gathered) or len(set(gathered)) > 1


def require_coherent(memory: str, group, device):
    """the coarse-grained fallback of exchange() (torch allocations + CUDA-IPC storage sharing) is only known to work between
    processes that share ONE GPU -- its L2 makes them coherent.  Across GPUs a spin on such memory can read a stale flag forever
    (csrc/peer_sync.h): every call would run into the wait bound and poison its output.  Raise instead, so that the callers' setup
    fails (`ok = False`, `why` says so) and they use RCCL.  Collective."""
    if memory.startswith("coarse-grained fallback") and spans_devices(group, device):
        raise RuntimeError("peer buffers fell back to coarse-grained memory (" + memory + ") and the group spans more than one GPU: "
                           "flag polling across devices needs uncached / fine-grained allocations; using the process group's collectives instead")

