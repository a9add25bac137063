"""Mirror of SURFGAN_3D/dataset.py for the training loop: NumpyPathDataset (path dataset over `*.npy`
volumes, shuffled sample buffer, batch / batch_mpi / repeat / split_by_fraction) and the numpy
normalisation helpers, plus a pinned-memory prefetcher that overlaps np.load + host-to-device copies with the
step (the reference loads synchronously on the training thread, optuna_objective.py:422-428).

batch_mpi(): the reference lets rank 0 draw the global batch and MPI-scatters the file lists
(dataset.py:307-333).  Here every rank keeps the SAME sample buffer (same shuffle seed) and takes its
column of the (-1, world) reshape, which is exactly what the scatter delivered: no collective on the data path.
"""
import copy
import glob
import os
import random
import shutil
import threading
import queue

import numpy as np


def normalize_numpy(unnormalized_input, mean, stddev, verbose=False):
    """dataset.py:78-97."""
    if mean is None and stddev is None:
        print("INFO: no data_mean or data_stddev was defined, not normalizing the input data")
        return unnormalized_input
    elif mean is None and stddev is not None:
        raise Exception("ERROR: data_stddev was defined, but data_mean was not. Either define both to apply input normalization, or define neither to not apply input normalization")
    elif mean is not None and stddev is None:
        raise Exception("ERROR: data_mean was defined, but data_stddev was not. Either define both to apply input normalization, or define neither to not apply input normalization")
    return (unnormalized_input - mean) / stddev


def invert_normalize_numpy(normalized_input, mean, stddev, verbose=False):
    """dataset.py:99-118."""
    if mean is None and stddev is None:
        print("INFO: no data_mean or data_stddev was defined, not inverting normalizing of the input data")
        return normalized_input
    elif mean is None and stddev is not None:
        raise Exception("ERROR: data_stddev was defined, but data_mean was not. Either define both to apply input normalization, or define neither to not apply input normalization")
    elif mean is not None and stddev is None:
        raise Exception("ERROR: data_mean was defined, but data_stddev was not. Either define both to apply input normalization, or define neither to not apply input normalization")
    return (normalized_input * stddev) + mean


class NumpyPathDataset:
    """dataset.py:155-349."""

    def __init__(self, npy_dir, scratch_dir, copy_files, is_correct_phase, rank=0, world_size=1, seed=None):
        self.npy_files = sorted(glob.glob(npy_dir + '*.npy'))   # sorted: every rank must see the same order
        print(f"Length of dataset: {len(self.npy_files)}")
        self.rank, self.world_size = rank, world_size
        self._rng = random.Random(seed) if seed is not None else random
        if scratch_dir is not None and scratch_dir[-1] == '/':
            scratch_dir = scratch_dir[:-1]
        if scratch_dir is None:
            self.scratch_dir = os.path.normpath(npy_dir)
            self.scratch_files = list(self.npy_files)
        else:
            self.scratch_dir = os.path.normpath(scratch_dir + npy_dir) if is_correct_phase else npy_dir
            self._copy_files_to_scratch(scratch_dir, copy_files, is_correct_phase)
            self.scratch_files = sorted(glob.glob(self.scratch_dir + '/*.npy'))
            assert len(self.scratch_files) == len(self.npy_files)
        self._init_samplebuffer()
        if len(self.scratch_files) > 0:
            test_npy_array = np.load(self.scratch_files[0], mmap_mode='r')[np.newaxis, ...]
            self.shape, self.dtype = test_npy_array.shape, test_npy_array.dtype
            del test_npy_array

    def _copy_files_to_scratch(self, scratch_dir, copy_files, is_correct_phase):
        if copy_files and is_correct_phase:
            os.makedirs(self.scratch_dir, exist_ok=True)
            for f in self.npy_files:
                if not os.path.isfile(os.path.normpath(scratch_dir + f)):
                    shutil.copy(f, os.path.normpath(scratch_dir + f))

    def _init_samplebuffer(self):
        self.samplebuffer = self.scratch_files[:]
        self._rng.shuffle(self.samplebuffer)

    def __iter__(self):
        for path in self.scratch_files:
            yield path

    def __getitem__(self, idx):
        return self.scratch_files[idx]

    def __len__(self):
        return len(self.scratch_files)

    def split_by_fraction(self, fraction):
        nsamples_dataset1 = int(np.round(fraction * len(self.scratch_files)) + 1e-5)
        nsamples_dataset2 = len(self.scratch_files)
        assert nsamples_dataset1 > 0 and nsamples_dataset2 > 0
        return self.split_by_index(nsamples_dataset1)

    def split_by_index(self, index):
        dataset1, dataset2 = copy.deepcopy(self), copy.deepcopy(self)
        dataset1.scratch_files, dataset2.scratch_files = self.scratch_files[0:index], self.scratch_files[index:]
        dataset1.npy_files, dataset2.npy_files = self.npy_files[0:index], self.npy_files[index:]
        dataset1._init_samplebuffer()
        dataset2._init_samplebuffer()
        return dataset1, dataset2

    def _load_batch_from_filelist(self, batch_paths):
        batch = [np.load(path).astype('float32') for path in batch_paths]
        if len(batch) > 0:
            batch = np.stack(batch)
            batch = batch[:, np.newaxis, ...]
        return batch

    def batch_paths(self, batch_size, auto_repeat=True):
        if batch_size > len(self.samplebuffer):
            if auto_repeat:
                self.repeat()
                return self.batch_paths(batch_size, auto_repeat)
            paths = self.samplebuffer
        else:
            paths = self.samplebuffer[0:batch_size]
            self.samplebuffer = self.samplebuffer[batch_size:]
        return paths

    def batch(self, batch_size, auto_repeat=True, verbose=False):
        paths = self.batch_paths(batch_size, auto_repeat)
        if verbose:
            print("Got batch:")
            for element in paths:
                print(element)
        return self._load_batch_from_filelist(paths)

    def batch_mpi_paths(self, batch_size, auto_repeat=True):
        world = self.world_size
        global_batch_size = batch_size * world
        if global_batch_size > len(self.samplebuffer):
            if auto_repeat:
                self.repeat()
                return self.batch_mpi_paths(batch_size, auto_repeat)
            paths = list(self.samplebuffer)
            while len(paths) % world > 0:
                paths.append(None)
        else:
            paths = self.samplebuffer[0:global_batch_size]
            self.samplebuffer = self.samplebuffer[global_batch_size:]
        mine = [paths[i] for i in range(self.rank, len(paths), world)]   # column `rank` of reshape(-1, world)
        while len(mine) > 0 and mine[-1] is None:
            mine.pop()
        return mine

    def batch_mpi(self, batch_size, auto_repeat=True, verbose=False):
        mine = self.batch_mpi_paths(batch_size, auto_repeat)
        if verbose:
            print(f"Worker: {self.rank}. Got batch: {mine}")
        return self._load_batch_from_filelist(mine)

    def repeat(self):
        new_samplebuffer = self.scratch_files[:]
        self._rng.shuffle(new_samplebuffer)
        self.samplebuffer.extend(new_samplebuffer)

    def print_samplebuffer(self):
        for path in self.samplebuffer:
            print(path)


class PinnedPrefetcher:
    """Background np.load -> normalise -> pinned buffer -> async H2D copy on a side stream, `depth` batches
    ahead.  next() returns a device tensor whose copy the current stream has been made to wait for."""

    def __init__(self, dataset, batch_size, distributed, mean=None, stddev=None, device='cuda', depth=2):
        import torch
        self.torch = torch
        self.ds, self.bs, self.dist = dataset, batch_size, distributed
        self.mean, self.std = mean, stddev
        self.device = torch.device(device)
        self.q = queue.Queue(maxsize=depth)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self._stop = False
        self.th = threading.Thread(target=self._work, daemon=True)
        self.th.start()

    def _work(self):
        torch = self.torch
        while not self._stop:
            paths = self.ds.batch_mpi_paths(self.bs) if self.dist else self.ds.batch_paths(self.bs)
            arr = self.ds._load_batch_from_filelist(paths)
            if self.mean is not None and self.std is not None:
                arr = (arr - np.float32(self.mean)) / np.float32(self.std)
            host = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
            if self.stream is not None:
                host = host.pin_memory()
                with torch.cuda.stream(self.stream):
                    dev = host.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                self.q.put((dev, ev, host))
            else:
                self.q.put((host, None, host))

    def next(self):
        dev, ev, _host = self.q.get()
        if ev is not None:
            self.torch.cuda.current_stream().wait_event(ev)
        return dev

    def close(self):
        self._stop = True
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
