"""Per-step host -> device parameter block.

Everything a training step's kernels read that changes from step to step and is decided on the host -- the RNG seeds
(SO(3) rotation, the two dropouts, the triplet sampler), the miner's per-class plan (prefix of triplet counts, triplets
per anchor) and RAdam's descriptor tables (step sizes) -- lives in ONE persistent device buffer with a fixed layout.
Each step the host fills a page-locked staging slot (a ring, so the host may run several steps ahead) and enqueues ONE
asynchronous copy into that buffer; the kernels take pointers into it.  Two things follow:

  * the step needs a single small upload instead of five, and no kernel argument changes from step to step, so
  * the whole step can be captured once as a hipGraph and replayed (hpcs_amd/step.py): addresses, grid sizes and scalar
    arguments are all constant, the per-step values arrive through this block.
"""
import struct

import torch

SEED_NAMES = ("rot", "drop1", "drop2", "miner")


def _pad(n, a=64):
    return (n + a - 1) // a * a


class StepParams:
    def __init__(self, device, num_class, radam_table_bytes, ring=6):
        self.device = torch.device(device)
        self.C = int(num_class)
        off = 0
        self.layout = {}

        def sec(name, nbytes):
            nonlocal off
            self.layout[name] = (off, nbytes)
            off = _pad(off + nbytes)
        sec("seeds", 8 * len(SEED_NAMES))
        sec("toff", 8 * (self.C + 1))
        sec("kk", 4 * self.C)
        for i, nb in enumerate(radam_table_bytes):
            sec(f"radam{i}", nb)
        self.nbytes = off
        self.dev = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.slots = [torch.zeros(self.nbytes, dtype=torch.uint8).pin_memory() for _ in range(ring)]
        self.events = [None] * ring
        self.i = 0
        self.cur = None
        o, _ = self.layout["seeds"]
        self.seed_dev = {n: self.dev[o + 8 * j:o + 8 * j + 8].view(torch.int64) for j, n in enumerate(SEED_NAMES)}
        o, nb = self.layout["toff"]
        self.toff_dev = self.dev[o:o + nb].view(torch.int64)
        o, nb = self.layout["kk"]
        self.kk_dev = self.dev[o:o + nb].view(torch.int32)
        self.radam_dev = []
        for i in range(len(radam_table_bytes)):
            o, nb = self.layout[f"radam{i}"]
            self.radam_dev.append(self.dev[o:o + nb])

    # ---- host side: fill a staging slot, then one async copy -------------------------------------------------------
    def begin(self):
        k = self.i
        self.i = (self.i + 1) % len(self.slots)
        if self.events[k] is not None:
            self.events[k].synchronize()          # normally long completed: the ring is several steps deep
        self.cur = k
        return self

    def put_seeds(self, **seeds):
        o, _ = self.layout["seeds"]
        buf = self.slots[self.cur]
        vals = [int(seeds.get(n, 0)) & (2 ** 63 - 1) for n in SEED_NAMES]
        buf[o:o + 8 * len(vals)] = torch.frombuffer(bytearray(struct.pack(f"<{len(vals)}q", *vals)), dtype=torch.uint8)

    def put_plan(self, plan):
        """plan: miner.loss_and_miner_utils.TripletPlan over exactly self.C classes."""
        if plan.C != self.C:
            raise ValueError(f"the sampling plan covers {plan.C} classes, the parameter block was sized for {self.C}")
        buf = self.slots[self.cur]
        o, nb = self.layout["toff"]
        buf[o:o + nb].view(torch.int64).copy_(plan.toff)
        o, nb = self.layout["kk"]
        buf[o:o + nb].view(torch.int32).copy_(plan.k)

    def put_radam(self, tables):
        """tables: [(lanes, host uint8 tensor, n_tensors, rows_total)] in the order the block was sized for."""
        buf = self.slots[self.cur]
        for i, (_, host, _, _) in enumerate(tables):
            o, nb = self.layout[f"radam{i}"]
            if host.numel() != nb:
                raise ValueError("RAdam table size changed (the parameter set must stay fixed)")
            buf[o:o + nb] = host

    def commit(self):
        """One host->device copy on the current stream (stream-ordered after the previous step's kernels)."""
        self.dev.copy_(self.slots[self.cur], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[self.cur] = ev
