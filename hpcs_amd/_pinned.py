"""Asynchronous host->device uploads of small per-step tables through a ring of pinned staging buffers.

`tensor.to(device)` from pageable memory blocks the host until the copy has executed, i.e. until everything queued
before it on the stream has finished -- one such call per step drains the launch queue.  The miner's per-class plan
and RAdam's descriptor table are a few KB each, so they go through page-locked ring slots with non_blocking copies;
a slot is reused only after the event recorded behind its last copy has completed.
"""
import torch


class PinnedRing:
    def __init__(self, slots=4):
        self.slots = [None] * slots
        self.events = [None] * slots
        self.i = 0

    def upload(self, host, device):
        """host: CPU tensor (any dtype, contiguous).  Returns a device tensor with the same dtype/shape."""
        host = host.contiguous()
        nbytes = host.numel() * host.element_size()
        k = self.i
        self.i = (self.i + 1) % len(self.slots)
        buf = self.slots[k]
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 4096), dtype=torch.uint8).pin_memory()
            self.slots[k] = buf
        elif self.events[k] is not None:
            self.events[k].synchronize()          # normally long completed: the ring is several steps deep
        view = buf[:nbytes].view(host.dtype).view(host.shape) if nbytes else buf[:0].view(host.dtype)
        if nbytes:
            view.copy_(host)
        dev = torch.empty(host.shape, dtype=host.dtype, device=device)
        if nbytes:
            dev.copy_(view, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return dev


_RINGS = {}


def upload(host, device, key="default"):
    ring = _RINGS.get(key)
    if ring is None:
        ring = _RINGS[key] = PinnedRing(8)
    return ring.upload(host, device)
