"""Device memory of the input pipeline: one arena per batch in flight, sized once, reused for ever.

Everything a prepared batch holds - voxels, coordinates, grid indices, rulebooks - is produced on the input
pipeline's side stream and consumed on the training stream one or two steps later, with ragged sizes that change from
batch to batch.  Through torch's caching allocator that is the worst case: a block freed by the training thread is held
back until the consuming stream has passed it (record_stream), the host runs two steps ahead of the GPU, and the side
stream's pool therefore keeps growing by hipMalloc for dozens of steps (round 3: 143 device allocations inside the
driver's 20 timed steps, each one a step 0.6-1.0 ms slower).  The reference never sees this problem because its voxels
come out of DataLoader workers as host arrays (pcdet/datasets/processor/data_processor.py:115-143).

Here a prepared batch owns a SLOT: a few large chunks carved by bump allocation (exact sizes, known after the plan's one
host round trip) plus named persistent buffers whose contents carry an invariant from one use to the next (the voxel
level's bitmap is all zero, the voxeliser's hash table is empty).  Slots rotate; the only synchronisation is one event per
slot: recorded on the training stream when the step that consumed the slot has been enqueued, waited for by the side stream
before the slot is written again.  No allocator traffic in steady state, no record_stream."""
import threading

import torch

_TLS = threading.local()
CHUNK_BYTES = 512 << 20


def current_slot():
    return getattr(_TLS, "slot", None)


class use_slot:
    """with use_slot(slot): ops.* index builders carve their outputs from `slot` (thread local)."""

    def __init__(self, slot):
        self.slot = slot

    def __enter__(self):
        self.prev = current_slot()
        _TLS.slot = self.slot
        return self.slot

    def __exit__(self, *exc):
        _TLS.slot = self.prev
        return False


class ArenaSlot:
    def __init__(self, device):
        self.device = torch.device(device)
        self.chunks = []            # uint8 tensors, never returned to the allocator
        self.cur, self.off = 0, 0
        self.persist = {}           # name -> (meta, tensor)
        self.state = {}             # small values carried between uses (the voxeliser's sticky per-sample capacity, ...)
        self.marks = {}             # name -> callable that puts persistent buffer `name` back into its clean state (enqueues a kernel)
        self.clean = set()          # persistent buffers known to be in their clean state
        self.free_event = None      # training-stream event behind the last consumer of this slot
        self.grown = 0              # chunks / persistent buffers allocated so far (steady state: stops changing)
        self.plans = 0              # plans built into this slot since reset() (a stage-2 pair holds two)

    def reset(self):
        """Start of a new use.  The clean-up kernels of the previous use (un-marking the voxel level's bitmap from the coordinate
        list that use left in this slot's chunks) are enqueued HERE, before anything can carve and overwrite those chunks."""
        for name, fn in list(self.marks.items()):
            fn()
            self.clean.add(name)
        self.marks.clear()
        self.cur, self.off, self.plans = 0, 0, 0

    def take(self, shape, dtype):
        if isinstance(shape, int):
            shape = (shape,)
        numel = 1
        for s in shape:
            numel *= int(s)
        item = torch.empty((), dtype=dtype).element_size()
        nbytes = max(numel * item, 1)
        need = (nbytes + 255) // 256 * 256
        while True:
            if self.cur < len(self.chunks):
                ch = self.chunks[self.cur]
                if self.off + need <= ch.numel():
                    view = ch[self.off:self.off + numel * item].view(dtype).view(*shape)
                    self.off += need
                    return view
                self.cur, self.off = self.cur + 1, 0
                continue
            self.chunks.append(torch.empty((max(need, CHUNK_BYTES),), dtype=torch.uint8, device=self.device))
            self.grown += 1

    def persistent(self, name, meta, nbytes, init=None):
        """A named buffer that survives reset().  Re-created (and `init` run on it) when `meta` changes.
        Returns (tensor, fresh)."""
        ent = self.persist.get(name)
        if ent is not None and ent[0] == meta and ent[1].numel() >= nbytes:
            return ent[1], False
        buf = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=self.device)
        self.grown += 1
        if init is not None:
            init(buf)
        self.persist[name] = (meta, buf)
        self.marks.pop(name, None)
        self.clean.discard(name)
        return buf, True


class IndexArena:
    """Round-robin slots for the batches an input pipeline has in flight (being prepared, prepared, being consumed)."""

    def __init__(self, device, slots=3):
        self.slots = [ArenaSlot(device) for _ in range(int(slots))]
        self.turn = 0

    def acquire(self, stream):
        """Next slot; `stream` (the side stream the slot is about to be written on) waits for its last consumer."""
        slot = self.slots[self.turn % len(self.slots)]
        self.turn += 1
        if slot.free_event is not None:
            stream.wait_event(slot.free_event)
        slot.reset()
        return slot

    @staticmethod
    def release(slot, stream):
        """Everything that reads `slot` has been enqueued on `stream`."""
        ev = slot.free_event
        if ev is None:
            ev = slot.free_event = torch.cuda.Event()
        ev.record(stream)

    @property
    def grown(self):
        return sum(s.grown for s in self.slots)
