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
slot: recorded on the training stream when the step that consumed the slot has been enqueued, queried on the host by the
pipeline's worker thread (or, host_wait=False, waited for by the side stream) before the slot is written again.  No
allocator traffic in steady state, no record_stream."""
import threading

import torch

_TLS = threading.local()
CHUNK_BYTES = 512 << 20
# How a pipeline's worker thread waits for an event on the host.  Blocking events + Event.synchronize(): the thread sleeps inside the
# runtime with the interpreter lock RELEASED.  The round-4 first form polled Event.query() every 50 us - 20,000 lock hand-overs a second
# taken from the thread that enqueues the step; harmless at 16 ms per step, but the forward-only loop (3.3 ms per step, 2.6 ms of them
# host work) fell into alternating 3.3 / 4.3 ms steps on some boxes (profiles/r04_bench_c2_step_ms.txt).  TODA_PREFETCH_POLL=1: poll.
import os as _os
BLOCKING_EVENTS = _os.environ.get("TODA_PREFETCH_POLL", "0") != "1"


def _host_wait(ev):
    if BLOCKING_EVENTS:
        ev.synchronize()
        return
    import time as _t
    while not ev.query():
        _t.sleep(5e-5)


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
        self.in_use = False         # handed to a consumer and not yet released

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
    """Slots for the batches an input pipeline has in flight (being prepared, prepared, being consumed).

    A slot is re-used when the event behind its last consumer has COMPLETED.  The check is made on the host (`Event.query`, by the
    pipeline's worker thread): with every slot in flight the worker sleeps until the oldest one is released and only then enqueues
    the next batch's index kernels, so the pipeline prepares one batch per GPU step, stays `slots - 2` steps ahead, and neither
    stream ever waits for the other on the GPU (`host_wait=False`: `stream.wait_event` instead).  `slots < max_slots`: grow instead of
    waiting.  After the first batch the other slots are laid out like the first (`prewarm`), so nothing is allocated once the first
    preparation is over."""

    def __init__(self, device, slots=4, max_slots=None, host_wait=True):
        self.device = torch.device(device)
        self.host_wait = bool(host_wait)
        max_slots = slots if max_slots is None else max_slots
        self.slots = [ArenaSlot(device) for _ in range(max(2, int(slots)))]
        self.max_slots = max(int(max_slots), len(self.slots))
        self.turn = 0
        self.waits = 0              # acquisitions that had to wait on the stream (max_slots reached)
        self._warm = False

    def acquire(self, stream):
        """A slot whose last consumer has finished (host-side check), else a new one; `stream` is the side stream it will be written on."""
        n = len(self.slots)
        pick = None
        for k in range(n):
            slot = self.slots[(self.turn + k) % n]
            if not slot.in_use and (slot.free_event is None or slot.free_event.query()):
                pick = slot
                self.turn = (self.turn + k + 1) % n
                break
        if pick is None and n < self.max_slots:
            pick = ArenaSlot(self.device)
            self.slots.append(pick)
            if self._warm:
                self._clone_layout(self.slots[0], pick)
        if pick is None:
            # every slot is in flight and the arena is at its bound: wait for the OLDEST release - on the host (poll + sleep; this is the
            # input pipeline's worker thread), so that the index kernels are enqueued only when they can run: the pipeline then prepares one
            # batch per GPU step and stays `max_slots - 2` steps ahead, with no queue-to-queue dependency on the GPU
            import time as _t
            cands = [sl for sl in self.slots if not sl.in_use and sl.free_event is not None] or [sl for sl in self.slots if not sl.in_use]
            if not cands:
                raise RuntimeError(f"IndexArena: all {n} slots are held by batches that were never released (a consumer keeps more than "
                                   f"{n - 1} prepared batches, or a preparation failed without abandon()); raise TODA_PREFETCH_MAX_SLOTS or release them")
            pick = cands[self.turn % len(cands)]
            self.turn = (self.turn + 1) % n
            if pick.free_event is not None:
                if self.host_wait:
                    _host_wait(pick.free_event)
                else:
                    stream.wait_event(pick.free_event)
            self.waits += 1
        pick.in_use = True          # until release(): a slot that is being prepared or consumed has no event to be judged by
        pick.reset()
        return pick

    def release(self, slot, stream):
        """Everything that reads `slot` has been enqueued on `stream`."""
        ev = slot.free_event
        if ev is None:
            ev = slot.free_event = torch.cuda.Event(blocking=BLOCKING_EVENTS)
        ev.record(stream)
        slot.in_use = False

    def abandon(self, slot):
        """A preparation into `slot` failed part-way: give the slot back.  Its persistent buffers may carry marks of the half-built batch
        (a bitmap with sites set, a hash table with entries) and no recorded way to undo them, so they are declared dirty - their next
        use clears them in full - and the pending un-mark callbacks, which would read coordinate lists that were never completed, are dropped."""
        slot.marks.clear()
        slot.clean.clear()
        slot.in_use = False

    def prewarm(self):
        """Lay every other slot out like the first one (chunk sizes, persistent buffers): called once, behind the first preparation."""
        if self._warm or not self.slots[0].chunks:
            return
        self._warm = True
        for slot in self.slots[1:]:
            if not slot.chunks:
                self._clone_layout(self.slots[0], slot)

    @staticmethod
    def _clone_layout(src, dst):
        for ch in src.chunks:
            dst.chunks.append(torch.empty_like(ch))
            dst.grown += 1
        for name, (meta, buf) in src.persist.items():
            # contents included: the voxeliser's hash table is in its clean state (every call leaves it so); a bitmap that carries marks
            # is not in dst.clean, so its first build clears it
            dst.persist[name] = (meta, buf.clone())
            dst.grown += 1
        dst.state.update({k: v for k, v in src.state.items()})

    @property
    def grown(self):
        return sum(s.grown for s in self.slots)
