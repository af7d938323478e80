"""The data loader of get_data_loader (reference randlanet/utils/dataset.py:100-131) with the clouds resident in
HBM and the per-item work of PointCloudPreprocessor.preprocess (dataset.py:61-97) done by one HIP kernel per
batch (rl_batch_assemble, include/rl_randlanet.h): sub-sample, optional normalisation, augmentation
(randlanet/utils/augmentation.py:147-167), float32 / int64 conversion and collation.

Iterating yields what the reference's DataLoader yields - (input (B,n,3+F) float32, labels (B,n) int64,
index (B,) int64) - with input / labels already on the device.

rng="numpy" (default): every random number is drawn on the host from numpy's and torch's GLOBAL generators, in
    the reference's order (the sampler's permutation like torch's RandomSampler, then per item: sample indices,
    jitter noise, scale, three angles, three shifts), so that a run seeded like a reference run sees the same
    batches (to float32 rounding) and leaves both generators where the reference leaves them.
rng="device": sample indices and jitter noise of a whole batch are drawn on the GPU by ONE launch (rl_batch_draw: a keyed
    permutation for the sample without replacement, Philox normals; a pure function of this loader's seed, the batch
    number, the cloud and the position) - the host cost of numpy's permutation + 3n normal draws, ~1.7 ms per
    40960-point cloud, caps the numpy mode near 580 clouds/s where this one feeds 870 (bench.py trainer_e2e); the seven
    per-cloud scalars still come from numpy.  Same distributions, different stream.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _hip as H
from . import preprocessing
from .augmentation import AugmentationSettings, _rotation
from .dataset import PointCloudPreprocessor

Sample = Tuple[np.ndarray, np.ndarray, np.ndarray]
_NORMALIZATION = {None: 0, "mean": 1, "max": 2, "stdev": 3}


class DeviceDataLoader:
    def __init__(self, dataset: Sequence[Sample], n_sample_points: int, batch_size: int, shuffle: bool = False,
                 consistent_sampling: bool = True, augmentation_settings: Optional[AugmentationSettings] = None,
                 normalization: Optional[str] = None, device=None, rng: str = "numpy") -> None:
        if rng not in ("numpy", "device"):
            raise ValueError(f"rng must be 'numpy' or 'device', got {rng!r}")
        self.dataset = PointCloudPreprocessor(dataset, n_sample_points, consistent_sampling=consistent_sampling,
                                              augmentation_settings=augmentation_settings, normalization=normalization)
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.rng = rng
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise H.HipKernelError("DeviceDataLoader needs a GPU (use get_data_loader for the host pipeline)")
        self._n = int(n_sample_points)
        self._consistent = bool(consistent_sampling)
        self._aug = augmentation_settings
        self._norm = _NORMALIZATION.get(normalization, 4)        # any other string: centre only (dataset.py:92)
        # device mode: ONE launch (rl_batch_draw) draws a batch's sample indices and jitter noise, a counter-based generator
        # keyed by (seed of this loader, batch number)
        self._seed = int(np.random.randint(0, 2 ** 31 - 1)) if rng == "device" else 0
        self._draws = 0
        self._consistent_idx = {}
        self._ring = None            # pinned host staging (see _staging)
        self._checks = []            # (pinned error words of a launched batch, event): see _check_launches
        # the whole dataset moves to HBM once
        self._xyz: List[torch.Tensor] = []
        self._feat: List[torch.Tensor] = []
        self._lab: List[torch.Tensor] = []
        self._F = None
        for xyz, features, labels in dataset:
            n = xyz.shape[0]
            assert xyz.ndim == 2 and xyz.shape[1] == 3, "Point coordinates should have shape (N, 3)!"
            assert features.shape[0] == n, "Features should have shape (N, F)!"
            assert labels.shape == (n,), "Labels should have shape (N,)!"
            if self._F is None:
                self._F = int(features.shape[1])
            assert features.shape[1] == self._F, "all clouds need the same number of features"
            xyz = np.ascontiguousarray(xyz if xyz.dtype == np.float64 else xyz.astype(np.float32, copy=False))
            self._xyz.append(torch.from_numpy(xyz).to(self.device))
            self._feat.append(torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32)).to(self.device))
            self._lab.append(torch.from_numpy(np.ascontiguousarray(labels).astype(np.int64)).to(self.device))
        self._F = self._F or 0

    def __len__(self) -> int:
        return (len(self._xyz) + self.batch_size - 1) // self.batch_size

    # ------------------------------------------------------------------------------ random draws
    def _order(self) -> List[int]:
        """Index order of one epoch, consuming torch's default generator like DataLoader + RandomSampler do."""
        torch.empty((), dtype=torch.int64).random_()                     # the DataLoader iterator's base seed
        if not self.shuffle:
            return list(range(len(self._xyz)))
        seed = int(torch.empty((), dtype=torch.int64).random_().item())  # RandomSampler.__iter__
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(len(self._xyz), generator=g).tolist()

    def _sample(self, n_src: int, host_row: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """Sample indices of one cloud on the device - or, numpy mode with `host_row` (a row of the pinned staging buffer):
        written there and None returned (the batch's rows go over in one asynchronous copy)."""
        n = self._n
        if self._consistent:
            key = n_src
            if key not in self._consistent_idx:    # seed-0 draw, the same every time (preprocessing.py:22-31)
                idx = preprocessing.sample_points(n_src, n, consistent=True)
                self._consistent_idx[key] = torch.from_numpy(idx.astype(np.int64)).to(self.device)
            return self._consistent_idx[key]
        if self.rng == "numpy":
            idx = preprocessing.sample_points(n_src, n, consistent=False)
            if host_row is not None:
                host_row.numpy()[:] = idx
                return None
            return torch.from_numpy(idx.astype(np.int64)).to(self.device)
        return None        # device mode: rl_batch_draw fills the batch's rows (see _assemble)

    def _job(self, cloud: int, job: H.CloudJob, host_noise: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """Fill the job record of one cloud; returns its jitter noise (n,3) float64 on the device, or None (no augmentation,
        or numpy mode with `host_noise`, a slice of the pinned staging buffer: the draw is written there)."""
        x = self._xyz[cloud]
        job.xyz, job.features, job.labels = x.data_ptr(), self._feat[cloud].data_ptr(), self._lab[cloud].data_ptr()
        job.n_points, job.xyz_f64 = x.shape[0], int(x.dtype == torch.float64)
        job.normalization, job.augment = self._norm, 0
        if not self._aug:
            return None
        a = self._aug
        job.augment = 1
        job.jitter_variance, job.jitter_limit = a.jitter_variance, a.jitter_limit
        # the reference's order of draws: jitter noise, scale, three angles, three shifts (augmentation.py:147-167)
        if self.rng == "numpy":
            draw = np.random.randn(self._n, 3)
            if host_noise is not None:
                host_noise.numpy()[:] = draw
                noise = None
            else:
                noise = torch.from_numpy(draw).to(self.device)
        else:
            noise = None           # device mode: rl_batch_draw
        job.scale = np.random.uniform(1 - a.scale_limit, 1 + a.scale_limit)
        assert len(a.rotation_angle_variances) == 3, "angle_sigmas should have length 3"
        assert len(a.rotation_angle_limits) == 3, "angle_clips should have length 3"
        angles = [float(np.clip(s * np.random.randn(), -lim, lim))
                  for s, lim in zip(a.rotation_angle_variances, a.rotation_angle_limits)]
        R = _rotation(*angles)
        shift = np.random.uniform(-a.shift_limit, a.shift_limit, 3)
        for i in range(9):
            job.R[i] = float(R.flat[i])
        for i in range(3):
            job.shift[i] = float(shift[i])
        return noise

    # ------------------------------------------------------------------------------ iteration
    def _staging(self, B: int):
        """Pinned host staging for one batch (job records; in the numpy mode also sample indices and jitter noise), from a
        small ring: everything the host draws reaches the device through asynchronous copies, so the host never waits for
        the GPU here (a pageable `.to(device)` is a synchronous copy - it stalled the training loop once per batch behind
        all the work queued on the stream).  A slot is reused only after the copies that last read it have executed."""
        if self._ring is None:
            n, Bmax = self._n, self.batch_size
            mk = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()
            self._ring = [dict(jobs=mk((Bmax * C.sizeof(H.CloudJob),), torch.uint8),
                               idx=mk((Bmax, n), torch.int64) if self.rng == "numpy" or self._consistent else None,
                               noise=mk((Bmax, n, 3), torch.float64) if (self.rng == "numpy" and self._aug) else None,
                               event=None) for _ in range(4)]
            self._slot = 0
        st = self._ring[self._slot]
        self._slot = (self._slot + 1) % len(self._ring)
        if st["event"] is not None:
            st["event"].synchronize()
        return st

    def _assemble(self, ids):
        """One batch on the CURRENT stream: the draws (in the reference's order), the staging copies, rl_batch_assemble."""
        n, F, dev = self._n, self._F, self.device
        B = len(ids)
        jobs = (H.CloudJob * B)()
        st = self._staging(B)
        host_idx = st["idx"] is not None and not self._consistent and self.rng == "numpy"
        host_noise = st["noise"] is not None
        indices = torch.empty((B, n), dtype=torch.int64, device=dev)
        noise = torch.empty((B, n, 3), dtype=torch.float64, device=dev) if self._aug else None
        for b, cloud in enumerate(ids):            # item by item, like DataLoader(num_workers=0)
            smp = self._sample(self._xyz[cloud].shape[0], st["idx"][b] if host_idx else None)
            if smp is not None:
                indices[b] = smp
            nz = self._job(cloud, jobs[b], st["noise"][b] if host_noise else None)
            if nz is not None:
                noise[b] = nz
        if host_idx:
            indices.copy_(st["idx"][:B], non_blocking=True)
        if host_noise:
            noise.copy_(st["noise"][:B], non_blocking=True)
        nbytes = B * C.sizeof(H.CloudJob)
        st["jobs"].numpy()[:nbytes] = np.frombuffer(bytes(jobs), dtype=np.uint8)
        jobs_dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        jobs_dev.copy_(st["jobs"][:nbytes], non_blocking=True)
        st["event"] = torch.cuda.Event()
        st["event"].record(torch.cuda.current_stream(dev))
        if self.rng == "device":
            want_idx = not self._consistent
            if want_idx or noise is not None:
                self._draws += 1
                H.check(H.lib().rl_batch_draw(jobs_dev.data_ptr(), B, n, (self._seed << 32) | (self._draws & 0xFFFFFFFF),
                                              indices.data_ptr() if want_idx else None, H.ptr(noise),
                                              torch.cuda.current_stream(dev).cuda_stream), "rl_batch_draw")
        scratch = torch.empty(H.lib().rl_batch_assemble_scratch_doubles(B, n), dtype=torch.float64, device=dev)
        inp = torch.empty((B, n, 3 + F), dtype=torch.float32, device=dev)
        lab = torch.empty((B, n), dtype=torch.int64, device=dev)
        H.check(H.lib().rl_batch_assemble(jobs_dev.data_ptr(), B, n, F, indices.data_ptr(), H.ptr(noise),
                                          scratch.data_ptr(), inp.data_ptr(), lab.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream), "rl_batch_assemble")
        # the launch's error words (a cloud-wide rendezvous that timed out, rl_randlanet.h) come back asynchronously and are
        # looked at when they have arrived - no host synchronisation on the way (the host runs ahead of the GPU)
        f0 = int(H.lib().rl_batch_assemble_flag_u32(B, n, 0))
        stride = int(H.lib().rl_batch_assemble_flag_u32(B, n, 1)) - f0 if B > 1 else 1
        words = torch.empty(B, dtype=torch.int32).pin_memory()
        words.copy_(scratch.view(torch.int32)[f0::stride][:B], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._checks.append((words, ev))
        self._check_launches(wait=False)
        return inp, lab, torch.tensor(ids, dtype=torch.int64)

    def _check_launches(self, wait: bool) -> None:
        """Raise if a finished rl_batch_assemble launch reported a timed-out rendezvous (its batch is not what the reference
        would have produced).  wait=False: only launches whose words have already arrived."""
        while self._checks:
            words, ev = self._checks[0]
            if wait:
                ev.synchronize()
            elif not ev.query():
                return
            self._checks.pop(0)
            if bool((words != 0).any()):
                raise H.HipKernelError("rl_batch_assemble: a cloud-wide rendezvous timed out (the launch was not co-resident: "
                                       "a CU mask or partition mode?); set RL_ASSEMBLE_ONE_WG=1")

    def __iter__(self):
        # (Assembling batches one ahead on a stream of the loader's own, beside the training step, was built and measured in
        # the device-rng mode: 762 vs 770 clouds/s through Model.train - the loader's sort / fill kernels, ~0.45 ms of GPU time per
        # step of four clouds, do not find room beside a replayed step graph.  Not kept.)
        order = self._order()
        for start in range(0, len(order), self.batch_size):
            yield self._assemble(order[start:start + self.batch_size])
        self._check_launches(wait=True)        # (end of an epoch: the trainer synchronises here anyway)


def get_device_data_loader(dataset: Sequence[Sample], n_sample_points: int, batch_size: int, shuffle: bool = False,
                           consistent_sampling: bool = True,
                           augmentation_settings: Optional[AugmentationSettings] = None,
                           normalization: Optional[str] = None, device=None, rng: str = "numpy") -> DeviceDataLoader:
    """Same arguments as get_data_loader (dataset.py:100-131) plus the device and the random-number source."""
    return DeviceDataLoader(dataset, n_sample_points, batch_size, shuffle=shuffle,
                            consistent_sampling=consistent_sampling, augmentation_settings=augmentation_settings,
                            normalization=normalization, device=device, rng=rng)
