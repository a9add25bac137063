"""Per-phase `.npy` volume source for the training loop: the interface of SURFGAN_3D/dataset.py
(`NumpyPathDataset.batch / batch_mpi / repeat / split_by_fraction`, `normalize_numpy`; reference dataset.py:78-118,
163-349, protocol self-test :357-395) on a design of its own:

* a dataset is a table of files plus an **index deck** (`_Deck`): a queue of integer positions into the table,
  reshuffled epoch by epoch from a private `random.Random(seed)`.  Drawing a batch pops positions; file names only
  appear when a batch is materialised.  `samplebuffer` is kept as a property (a path view of the deck) because the
  reference's callers and its self-test inspect it.
* `batch_mpi()`: the reference lets rank 0 draw `batch * world` paths and MPI-scatters the columns of the
  `(-1, world)` reshape (dataset.py:307-333).  Here every rank owns the same deck (same seed, sorted table) and
  takes column `rank` itself: same partition of every epoch, no collective on the data path.
* scratch staging (dataset.py:163-200): ONE process per node (`local_rank == 0`) copies, through a temporary name
  and an atomic rename so a half-written file is never visible under its final name; every process then waits until
  the whole table is present (the reference's busy-wait, dataset.py:176-177).
* `PinnedPrefetcher`: a ring of preallocated pinned host buffers and device buffers.  A worker thread fills slot
  after slot (np.load straight into the pinned buffer, normalise in place, `hipMemcpyAsync` on a side stream); a
  slot is reused only after an event recorded on the CONSUMER's stream, once the consumer has moved on to the next
  batch, has completed -- the step that reads a batch can never see it overwritten.
"""
import glob
import os
import queue
import random
import shutil
import threading
import time

import numpy as np


def _norm_args(mean, stddev, what):
    """Both given -> True, neither -> False (with the reference's notice), one of them -> error (dataset.py:87-95)."""
    if (mean is None) != (stddev is None):
        given, missing = ('data_stddev', 'data_mean') if mean is None else ('data_mean', 'data_stddev')
        raise Exception(f"ERROR: {given} was defined, but {missing} was not. Either define both to apply input "
                        f"normalization, or define neither to not apply input normalization")
    if mean is None:
        print(f"INFO: no data_mean or data_stddev was defined, not {what} of the input data")
        return False
    return True


def normalize_numpy(unnormalized_input, mean, stddev, verbose=False):
    """(x - mean) / stddev, or x unchanged when neither is given (reference dataset.py:78-97)."""
    if not _norm_args(mean, stddev, 'normalizing'):
        return unnormalized_input
    return (unnormalized_input - mean) / stddev


def invert_normalize_numpy(normalized_input, mean, stddev, verbose=False):
    """x * stddev + mean (reference dataset.py:99-118)."""
    if not _norm_args(mean, stddev, 'inverting normalizing'):
        return normalized_input
    return (normalized_input * stddev) + mean


class _Deck:
    """Queue of table positions; `refill()` appends one freshly shuffled epoch (reference `repeat()`)."""

    def __init__(self, size, rng):
        self.size, self.rng = size, rng
        self.cards = []
        self.refill()

    def refill(self):
        epoch = list(range(self.size))
        self.rng.shuffle(epoch)
        self.cards.extend(epoch)

    def draw(self, count, refill):
        """Pops `count` positions.  Short deck: refill until it is long enough, or (refill=False) hand out the rest
        WITHOUT consuming it, as the reference does (dataset.py:276-278)."""
        while count > len(self.cards):
            if not refill or self.size == 0:
                return list(self.cards)
            self.refill()
        out, self.cards = self.cards[:count], self.cards[count:]
        return out


class NumpyPathDataset:
    """Files `npy_dir*.npy` (optionally staged under `scratch_dir`), drawn epoch-wise in shuffled order."""

    def __init__(self, npy_dir, scratch_dir, copy_files, is_correct_phase, rank=0, world_size=1, seed=None,
                 stage_timeout=3600.0):
        self.rank, self.world_size = int(rank), int(world_size)
        self._rng = random.Random(seed) if seed is not None else random.Random()
        self.npy_files = sorted(glob.glob(npy_dir + '*.npy'))   # sorted: every rank must hold the same table
        print(f"Length of dataset: {len(self.npy_files)}")
        staged = scratch_dir is not None and is_correct_phase
        if staged:
            root = scratch_dir.rstrip('/')
            self.scratch_dir = os.path.normpath(root + npy_dir)
            self.scratch_files = [os.path.normpath(root + f) for f in self.npy_files]
            if copy_files:
                self._stage(self.npy_files, self.scratch_files)
            self._await_files(self.scratch_files, stage_timeout)
        else:
            self.scratch_dir = os.path.normpath(npy_dir)
            self.scratch_files = list(self.npy_files)
        self._deck = _Deck(len(self.scratch_files), self._rng)
        self.shape = self.dtype = None
        if self.scratch_files:
            probe = np.load(self.scratch_files[0], mmap_mode='r')
            self.shape, self.dtype = (1,) + tuple(probe.shape), probe.dtype
            del probe

    # ---- scratch staging ----------------------------------------------------------------------------------
    @staticmethod
    def _stage(sources, targets):
        for src, dst in zip(sources, targets):
            if os.path.isfile(dst):
                continue
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            tmp = f'{dst}.part{os.getpid()}'
            shutil.copy(src, tmp)
            os.replace(tmp, dst)       # atomic: readers see the whole file or none

    @staticmethod
    def _await_files(paths, timeout):
        deadline = time.time() + timeout
        missing = list(paths)
        while missing:
            missing = [p for p in missing if not os.path.isfile(p)]
            if missing:
                if time.time() > deadline:
                    raise TimeoutError(f'{len(missing)} file(s) never reached the scratch directory, e.g. {missing[0]}')
                time.sleep(0.2)

    # ---- the table ----------------------------------------------------------------------------------------
    def __iter__(self):
        return iter(self.scratch_files)

    def __getitem__(self, idx):
        return self.scratch_files[idx]

    def __len__(self):
        return len(self.scratch_files)

    def _subset(self, lo, hi):
        part = object.__new__(NumpyPathDataset)
        part.__dict__.update(self.__dict__)
        part.npy_files, part.scratch_files = self.npy_files[lo:hi], self.scratch_files[lo:hi]
        # children shuffle independently of the parent -- and WITHOUT advancing its generator: the order of the training
        # batches (drawn from the full set: quirk Q3) must not depend on whether a validation split was made
        probe = random.Random()
        probe.setstate(self._rng.getstate())
        part._rng = random.Random(f'{probe.random()!r}:{lo}:{hi}')
        part._deck = _Deck(len(part.scratch_files), part._rng)
        return part

    def split_by_index(self, index):
        """Two datasets: table rows [0, index) and [index, end)."""
        return self._subset(0, index), self._subset(index, len(self.scratch_files))

    def split_by_fraction(self, fraction):
        """E.g. 0.8 -> (first 80 % of the table, the rest), rounded to the nearest sample (dataset.py:218-233)."""
        first = int(np.round(fraction * len(self.scratch_files)) + 1e-5)
        assert first > 0 and len(self.scratch_files) > 0
        return self.split_by_index(first)

    # ---- the deck -----------------------------------------------------------------------------------------
    @property
    def samplebuffer(self):
        return [self.scratch_files[i] for i in self._deck.cards]

    @samplebuffer.setter
    def samplebuffer(self, paths):
        pos = {p: i for i, p in enumerate(self.scratch_files)}
        self._deck.cards = [pos[p] for p in paths]

    def repeat(self):
        self._deck.refill()

    def print_samplebuffer(self):
        for path in self.samplebuffer:
            print(path)

    def batch_paths(self, batch_size, auto_repeat=True):
        return [self.scratch_files[i] for i in self._deck.draw(batch_size, auto_repeat)]

    def batch_mpi_paths(self, batch_size, auto_repeat=True):
        """This rank's share of the next global batch: entries rank, rank + world, ... of the drawn positions (column
        `rank` of the reference's reshape(-1, world) before its scatter; a short last draw gives the low ranks one
        more sample, which is what padding with None and stripping it per rank did, dataset.py:316-338)."""
        drawn = self._deck.draw(batch_size * self.world_size, auto_repeat)
        return [self.scratch_files[i] for i in drawn[self.rank::self.world_size]]

    # ---- materialising ------------------------------------------------------------------------------------
    def load_into(self, paths, out):
        """np.load every path into out[i, 0] (float32, channel axis added); `out` may be pinned memory."""
        for i, p in enumerate(paths):
            np.copyto(out[i, 0], np.load(p), casting='unsafe')
        return out

    def _load_batch_from_filelist(self, batch_paths):
        """[N, 1, Z, Y, X] float32; an empty list stays an empty list (dataset.py:254-262)."""
        if len(batch_paths) == 0:
            return []
        first = np.load(batch_paths[0])
        out = np.empty((len(batch_paths), 1) + first.shape, dtype=np.float32)
        out[0, 0] = first
        self.load_into(batch_paths[1:], out[1:])
        return out

    def batch(self, batch_size, auto_repeat=True, verbose=False):
        paths = self.batch_paths(batch_size, auto_repeat)
        if verbose:
            print("Got batch:")
            for p in paths:
                print(p)
        return self._load_batch_from_filelist(paths)

    def batch_mpi(self, batch_size, auto_repeat=True, verbose=False):
        paths = self.batch_mpi_paths(batch_size, auto_repeat)
        if verbose:
            print(f"Worker: {self.rank}. Got batch: {paths}")
        return self._load_batch_from_filelist(paths)


class _Slot:
    __slots__ = ('host', 'dev', 'ready', 'released')


class PinnedPrefetcher:
    """Keeps `depth` batches in flight: np.load -> pinned buffer (normalised in place) -> async copy on a side stream.

    next() makes the caller's current stream wait for the copy and returns the slot's device tensor.  The tensor stays
    valid until the call AFTER the next one: calling next() again records, on the caller's stream, the event that
    releases the previous slot, and the worker does not touch a slot before that event has completed."""

    def __init__(self, dataset, batch_size, distributed, mean=None, stddev=None, device='cuda', depth=2):
        import torch
        self.torch = torch
        self.ds, self.bs, self.dist = dataset, int(batch_size), bool(distributed)
        self.normalise = _norm_args(mean, stddev, 'normalizing') if (mean is not None or stddev is not None) else False
        self.mean, self.std = (np.float32(mean), np.float32(stddev)) if self.normalise else (None, None)
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.nslots = depth + 2                       # `depth` queued + one at the consumer + one being filled
        self.slots = []
        self.free = queue.Queue()
        self.full = queue.Queue(maxsize=depth)
        self.held = None                              # slot index the consumer is reading
        self.error = None
        self._stop = threading.Event()
        for i in range(self.nslots):
            self.slots.append(None)
            self.free.put(i)
        self.th = threading.Thread(target=self._work, name='saragan-prefetch', daemon=True)
        self.th.start()

    def _make_slot(self, shape):
        torch = self.torch
        s = _Slot()
        s.host = torch.empty(shape, dtype=torch.float32, pin_memory=self.cuda)
        s.dev = torch.empty(shape, dtype=torch.float32, device=self.device) if self.cuda else s.host
        s.ready = torch.cuda.Event() if self.cuda else None
        s.released = None
        return s

    def _work(self):
        torch = self.torch
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            while not self._stop.is_set():
                try:
                    i = self.free.get(timeout=0.1)
                except queue.Empty:
                    continue
                paths = self.ds.batch_mpi_paths(self.bs) if self.dist else self.ds.batch_paths(self.bs)
                shape = (len(paths),) + tuple(self.ds.shape)
                slot = self.slots[i]
                if slot is None or tuple(slot.host.shape) != shape:
                    slot = self.slots[i] = self._make_slot(shape)
                if slot.released is not None:
                    slot.released.synchronize()       # the consumer's kernels that read this slot are done
                    slot.released = None
                host = slot.host.numpy()
                self.ds.load_into(paths, host)
                if self.normalise:
                    np.subtract(host, self.mean, out=host)
                    np.divide(host, self.std, out=host)
                if self.cuda:
                    with torch.cuda.stream(self.stream):
                        slot.dev.copy_(slot.host, non_blocking=True)
                        slot.ready.record(self.stream)
                while not self._stop.is_set():
                    try:
                        self.full.put(i, timeout=0.1)
                        break
                    except queue.Full:
                        continue
        except BaseException as e:   # surfaced by next(): a dead loader must not look like a slow one
            self.error = e

    def _release_held(self):
        if self.held is None:
            return
        slot = self.slots[self.held]
        if self.cuda:
            slot.released = self.torch.cuda.Event()
            slot.released.record(self.torch.cuda.current_stream(self.device))
        self.free.put(self.held)
        self.held = None

    def next(self):
        self._release_held()
        while True:
            if self.error is not None:
                raise RuntimeError('prefetch worker failed') from self.error
            try:
                i = self.full.get(timeout=0.5)
                break
            except queue.Empty:
                if not self.th.is_alive() and self.error is None:
                    raise RuntimeError('prefetch worker exited')
        slot = self.slots[i]
        if self.cuda:
            self.torch.cuda.current_stream(self.device).wait_event(slot.ready)
        self.held = i
        return slot.dev

    def close(self):
        self._stop.set()
        self.th.join(timeout=10.0)
        if self.cuda:
            self.torch.cuda.current_stream(self.device).synchronize()
        self.slots = []
