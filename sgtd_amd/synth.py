"""Deterministic synthetic keypoint / map generator.

The reference's datasets (semantic-graph JSONs of MulRan / MCD / Apollo scans,
README.md:45) are external downloads and not available offline, so benchmarks
and parity tests run on synthetic *semantic keypoint* frames with the same
layout the hot path consumes: per frame N instance centroids (xyz f32, sensor
frame) + a class label per centroid (labels 3..11 as the reference's graph
producer emits, src/sgtd/src/get_json.cpp:10-12,287-293; 0..12 for the "wild"
mapping, get_json_wild.cpp:10-12).

World model: static landmarks with density rho = N/(pi*50^2) on a square
region; a closed Lissajous trajectory with poses every `spacing` metres folded
so that (trajectory length * 100 m swath) ~= region area; map frame k observes
the N landmarks nearest to pose k, expressed in the sensor frame (yaw =
heading) with N(0, sigma_map) noise; a query re-observes the N landmarks
nearest to a perturbed copy of a random map pose (uniform yaw, N(0,0.5 m)
shift) with N(0, sigma_query) noise, ground truth = that map frame.
All randomness: numpy PCG64 seeded with (20251121, stream).
"""
import os
from dataclasses import dataclass

import numpy as np

BASE_SEED = 20251121


@dataclass
class SynthMap:
    xyz: np.ndarray        # (F, N, 3) float32, sensor frame
    label: np.ndarray      # (F, N) uint32
    pose: np.ndarray       # (F, 3) x, y, yaw
    landmarks: np.ndarray  # (L, 3) float64 world
    landmark_label: np.ndarray  # (L,) uint32


@dataclass
class SynthQueries:
    xyz: np.ndarray     # (Q, N, 3) float32
    label: np.ndarray   # (Q, N) uint32
    gt_frame: np.ndarray  # (Q,) int64 — the map frame each query re-observes
    pose: np.ndarray    # (Q, 3)


def _rng(stream, sub=0):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([BASE_SEED, stream, sub])))


def _trajectory(n_frames, spacing, swath):
    """closed Lissajous resampled at equal arc length `spacing`"""
    length = n_frames * spacing
    side = max(np.sqrt(length * swath), 2.2 * swath)
    amp = side / 2.0
    t = np.linspace(0.0, 2 * np.pi, 200001)

    def curve(p, a):
        return a * np.sin(p * t), a * np.sin((p + 1) * t + np.pi / 2)

    def arclen(p, a):
        x, y = curve(p, a)
        return np.sum(np.hypot(np.diff(x), np.diff(y)))

    p = 1
    while arclen(p, amp) < length and p < 4096:
        p += 1
    amp = amp * length / arclen(p, amp)  # arc length is linear in the amplitude
    x, y = curve(p, amp)
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])
    target = np.arange(n_frames) * spacing
    px = np.interp(target, s, x)
    py = np.interp(target, s, y)
    ahead = np.minimum(target + 0.5, s[-1])
    yaw = np.arctan2(np.interp(ahead, s, y) - py, np.interp(ahead, s, x) - px)
    return np.stack([px, py, yaw], axis=1), amp


def _observe(landmarks, labels, tree, pose, n_kp, sigma, rng, tie_k=10):
    """the n_kp landmarks nearest to each pose, in the sensor frame, noisy, f32, shuffled"""
    _, idx = tree.query(pose[:, :2], k=n_kp)
    idx = np.asarray(idx).reshape(pose.shape[0], n_kp)
    perm = np.argsort(rng.random(idx.shape), axis=1)       # seeded shuffle per frame
    idx = np.take_along_axis(idx, perm, axis=1)
    pts = landmarks[idx]                                    # (F, N, 3) world
    d = pts[:, :, :2] - pose[:, None, :2]
    c, s = np.cos(-pose[:, 2])[:, None], np.sin(-pose[:, 2])[:, None]
    out = np.empty_like(pts)
    out[:, :, 0] = c * d[:, :, 0] - s * d[:, :, 1]
    out[:, :, 1] = s * d[:, :, 0] + c * d[:, :, 1]
    out[:, :, 2] = pts[:, :, 2]
    clean = out
    out = (clean + rng.normal(0.0, sigma, size=clean.shape)).astype(np.float32)
    # frames with duplicate points or exact f32 k-NN distance ties are regenerated
    # (FLANN's tie order is unpinned): fresh noise from a sub-stream until the frame is tie-free
    if tie_k:
        bad = frames_with_knn_ties(out, tie_k)
        attempt = 0
        while bad.size:
            attempt += 1
            if attempt > 16:
                raise RuntimeError("could not draw tie-free frames")
            sub = np.random.Generator(np.random.PCG64(np.random.SeedSequence([BASE_SEED, 7919, attempt, int(bad[0]), int(bad.size)])))
            out[bad] = (clean[bad] + sub.normal(0.0, sigma, size=clean[bad].shape)).astype(np.float32)
            bad = bad[frames_with_knn_ties(out[bad], tie_k)]
    return out, labels[idx].astype(np.uint32)


GEN_VERSION = 2     # bump whenever _trajectory, _observe or the landmark model change: cached maps of older code are then ignored


def _cache_path(kind, key):
    """generated sets are deterministic functions of their parameters: large ones are kept under
    SGTD_SYNTH_CACHE (default /tmp/sgtd_synth_cache; empty = off) so that the test session and the
    bench that follows it on the same box do not regenerate the same 10 000-frame map several times"""
    import hashlib
    root = os.environ.get("SGTD_SYNTH_CACHE", "/tmp/sgtd_synth_cache")
    if not root:
        return None
    import scipy
    # (the generator's own version and scipy's: cKDTree's tie order and any change to _trajectory / _observe give other frames)
    h = hashlib.sha256(repr((kind, BASE_SEED, GEN_VERSION, key, np.__version__, scipy.__version__)).encode()).hexdigest()[:24]
    return os.path.join(root, "%s_%s.npz" % (kind, h))


def make_map(n_frames, n_kp=200, stream=1, spacing=2.0, swath=100.0, radius=50.0,
             sigma=0.02, label_lo=3, label_hi=11, z_sigma=1.5):
    from scipy.spatial import cKDTree

    cache = _cache_path("map", (n_frames, n_kp, stream, spacing, swath, radius, sigma, label_lo, label_hi, z_sigma)) if n_frames >= 2000 else None
    if cache and os.path.exists(cache):
        try:
            z = np.load(cache)
            m = SynthMap(xyz=z["xyz"], label=z["label"], pose=z["pose"], landmarks=z["landmarks"], landmark_label=z["landmark_label"])
            m._tree = cKDTree(m.landmarks[:, :2])
            return m
        except Exception:
            pass      # a damaged file: regenerate

    rng = _rng(stream, 0)
    pose, amp = _trajectory(n_frames, spacing, swath)
    rho = n_kp / (np.pi * radius * radius)
    half = amp + 2.5 * radius
    n_land = int(rho * (2 * half) ** 2)
    landmarks = np.empty((n_land, 3))
    landmarks[:, 0] = rng.uniform(-half, half, n_land)
    landmarks[:, 1] = rng.uniform(-half, half, n_land)
    landmarks[:, 2] = rng.normal(0.0, z_sigma, n_land)
    lab = rng.integers(label_lo, label_hi + 1, n_land).astype(np.uint32)
    tree = cKDTree(landmarks[:, :2])
    xyz, label = _observe(landmarks, lab, tree, pose, n_kp, sigma, rng)
    m = SynthMap(xyz=xyz, label=label, pose=pose, landmarks=landmarks, landmark_label=lab)
    m._tree = tree
    if cache:
        try:
            os.makedirs(os.path.dirname(cache), exist_ok=True)
            tmp = cache + ".%d.tmp.npz" % os.getpid()
            np.savez(tmp, xyz=xyz, label=label, pose=pose, landmarks=landmarks, landmark_label=lab)
            os.replace(tmp, cache)
        except Exception:
            pass
    return m


def make_queries(smap, n_queries, stream=1, sigma=0.05, shift_sigma=0.5, frames=None):
    """queries re-observing random (or the given) map frames from perturbed poses"""
    rng = _rng(stream, 1)
    n_frames = smap.pose.shape[0]
    if frames is None:
        gt = rng.integers(0, n_frames, n_queries)
    else:
        gt = np.asarray(frames, dtype=np.int64)
        n_queries = len(gt)
    pose = smap.pose[gt].copy()
    pose[:, :2] += rng.normal(0.0, shift_sigma, (n_queries, 2))
    pose[:, 2] = rng.uniform(-np.pi, np.pi, n_queries)
    n_kp = smap.xyz.shape[1]
    xyz, label = _observe(smap.landmarks, smap.landmark_label, smap._tree, pose, n_kp, sigma, rng)
    return SynthQueries(xyz=xyz, label=label, gt_frame=gt.astype(np.int64), pose=pose)


def effective_cpus():
    """host threads this process may really use: the affinity mask, capped by the cgroup CPU quota
    (a container that shows 256 hardware threads may be allowed 16 CPUs' worth of time: teams sized
    by the former only get in each other's way)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def frames_with_knn_ties(xyz, k, chunk=512):
    """indices of the frames of xyz (F, N, 3) f32 that have duplicate points or an exact f32
    distance tie among the k+1 nearest of any point (the same test as has_knn_ties, batched;
    torch CPU kernels: individually rounded IEEE f32 operations, all host threads)"""
    import torch
    if torch.get_num_threads() > effective_cpus():
        torch.set_num_threads(effective_cpus())
    x = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float32))
    kk = min(k + 1, x.shape[1])
    bad = []
    for f0 in range(0, x.shape[0], chunk):
        c = x[f0:f0 + chunk]
        dx = c[:, :, None, 0] - c[:, None, :, 0]
        dy = c[:, :, None, 1] - c[:, None, :, 1]
        dz = c[:, :, None, 2] - c[:, None, :, 2]
        d2 = (dx * dx + dy * dy) + dz * dz
        part = torch.topk(d2, kk, dim=2, largest=False, sorted=True).values
        tie = (part[:, :, 1:] == part[:, :, :-1]).any(dim=2).any(dim=1)
        bad.append(f0 + torch.nonzero(tie)[:, 0].numpy())
    return np.concatenate(bad) if bad else np.zeros(0, np.int64)


def has_knn_ties(xyz, k):
    """True if a frame has duplicate points or an exact f32 distance tie among
    the k+1 nearest of any point (FLANN's tie order is unpinned: such frames
    are excluded from golden fixtures)."""
    x = np.asarray(xyz, dtype=np.float32)
    d = x[:, None, :] - x[None, :, :]
    d2 = (d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]) + d[:, :, 2] * d[:, :, 2]
    part = np.sort(d2, axis=1)[:, :k + 1]
    return bool(np.any(part[:, 1:] == part[:, :-1]))


# ---------------------------------------------------------------------------
# A skewed, reference-shaped workload: what the reference's producers emit differs from the
# uniform generator above in three ways that all lengthen buckets — class frequencies are heavily skewed
# (get_json.cpp:10-12,287-293 maps SemanticKITTI classes to node labels, a handful of classes carry most instances;
# the wild mapping get_json_wild.cpp:10-12 has 13 classes), the number of instances per scan varies by almost an
# order of magnitude, and instances come in clusters (parked cars, tree rows, poles along a road).
# ---------------------------------------------------------------------------
@dataclass
class RaggedFrames:
    xyz: np.ndarray        # (total, 3) float32, sensor frame, frame after frame
    label: np.ndarray      # (total,) uint32
    kp_off: np.ndarray     # (F + 1,) int64
    pose: np.ndarray       # (F, 3) x, y, yaw
    gt_frame: np.ndarray   # (F,) int64 (queries: the map frame re-observed; maps: arange)

    @property
    def n_frames(self):
        return len(self.kp_off) - 1

    def frame(self, f):
        a, b = int(self.kp_off[f]), int(self.kp_off[f + 1])
        return self.xyz[a:b], self.label[a:b]

    def take(self, frames):
        """the given frames as a new set (offsets rebuilt)"""
        frames = np.asarray(frames, np.int64)
        n = (self.kp_off[frames + 1] - self.kp_off[frames]).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
        idx = np.concatenate([np.arange(self.kp_off[f], self.kp_off[f + 1]) for f in frames]) if len(frames) else np.zeros(0, np.int64)
        return RaggedFrames(self.xyz[idx], self.label[idx], off, self.pose[frames], self.gt_frame[frames])


@dataclass
class SkewWorld:
    landmarks: np.ndarray       # (L, 3) world
    landmark_label: np.ndarray  # (L,) uint32
    pose: np.ndarray            # (F, 3) map poses


def zipf_class_probs(n_classes=13, s=1.2):
    p = 1.0 / np.arange(1, n_classes + 1) ** s
    return p / p.sum()


def _observe_ragged(landmarks, labels, tree, pose, n_kp, sigma, rng, tie_k=10):
    """frame i = the n_kp[i] landmarks nearest to pose i (sensor frame, noisy, f32, shuffled); frames with duplicate
    points or exact f32 k-NN distance ties are redrawn like _observe's"""
    n_kp = np.asarray(n_kp, np.int64)
    F = len(n_kp)
    out_xyz, out_lab = [None] * F, [None] * F
    for n in np.unique(n_kp):
        sel = np.nonzero(n_kp == n)[0]
        sub = np.random.Generator(np.random.PCG64(np.random.SeedSequence([BASE_SEED, 104729, int(n), int(rng.integers(1 << 30))])))
        x, l = _observe(landmarks, labels, tree, pose[sel], int(n), sigma, sub, tie_k=tie_k if n > tie_k else 0)
        for j, f in enumerate(sel):
            out_xyz[f], out_lab[f] = x[j], l[j]
    off = np.concatenate([[0], np.cumsum(n_kp)]).astype(np.int64)
    return np.concatenate(out_xyz).astype(np.float32), np.concatenate(out_lab).astype(np.uint32), off


def make_skewed_map(n_frames, stream=1, kp_lo=50, kp_hi=400, n_classes=13, zipf_s=1.2, spacing=2.0, swath=100.0,
                    radius=50.0, sigma=0.02, z_sigma=1.5, cluster_frac=0.5, cluster_mean=6.0, cluster_sigma=15.0):
    """-> (RaggedFrames map, SkewWorld).  Labels Zipf-distributed over `n_classes` classes (label = class index, the most
    frequent first), kp_lo..kp_hi keypoints per frame (uniform), landmarks clustered: `cluster_frac` of them in
    Gaussian clusters of `cluster_mean` members on average (sigma `cluster_sigma` m; a cluster's members share its class
    with probability 0.8), the rest uniform.  Mean density = the uniform generator's ((kp_lo + kp_hi) / 2 in a
    `radius` view)."""
    from scipy.spatial import cKDTree
    rng = _rng(stream, 40)
    pose, amp = _trajectory(n_frames, spacing, swath)
    mean_kp = 0.5 * (kp_lo + kp_hi)
    rho = mean_kp / (np.pi * radius * radius)
    half = amp + 2.5 * radius
    n_land = int(rho * (2 * half) ** 2)
    n_cl_members = int(cluster_frac * n_land)
    n_clusters = max(1, int(n_cl_members / cluster_mean))
    probs = zipf_class_probs(n_classes, zipf_s)
    centre = rng.uniform(-half, half, (n_clusters, 2))
    centre_class = rng.choice(n_classes, n_clusters, p=probs)
    member_of = rng.integers(0, n_clusters, n_cl_members)
    xy_c = centre[member_of] + rng.normal(0.0, cluster_sigma, (n_cl_members, 2))
    own = rng.random(n_cl_members) < 0.8
    lab_c = np.where(own, centre_class[member_of], rng.choice(n_classes, n_cl_members, p=probs))
    n_bg = n_land - n_cl_members
    xy_b = rng.uniform(-half, half, (n_bg, 2))
    lab_b = rng.choice(n_classes, n_bg, p=probs)
    landmarks = np.empty((n_land, 3))
    landmarks[:, :2] = np.concatenate([xy_c, xy_b])
    landmarks[:, 2] = rng.normal(0.0, z_sigma, n_land)
    lab = np.concatenate([lab_c, lab_b]).astype(np.uint32)
    tree = cKDTree(landmarks[:, :2])
    n_kp = rng.integers(kp_lo, kp_hi + 1, n_frames)
    xyz, label, off = _observe_ragged(landmarks, lab, tree, pose, n_kp, sigma, rng)
    world = SkewWorld(landmarks=landmarks, landmark_label=lab, pose=pose)
    world._tree = tree
    world._kp = (kp_lo, kp_hi)
    return RaggedFrames(xyz, label, off, pose, np.arange(n_frames, dtype=np.int64)), world


def make_skewed_queries(world, n_queries, stream=1, sigma=0.05, shift_sigma=0.5):
    """queries re-observing random map poses of a skewed world from perturbed poses, each with its own keypoint count"""
    rng = _rng(stream, 41)
    gt = rng.integers(0, world.pose.shape[0], n_queries)
    pose = world.pose[gt].copy()
    pose[:, :2] += rng.normal(0.0, shift_sigma, (n_queries, 2))
    pose[:, 2] = rng.uniform(-np.pi, np.pi, n_queries)
    n_kp = rng.integers(world._kp[0], world._kp[1] + 1, n_queries)
    xyz, label, off = _observe_ragged(world.landmarks, world.landmark_label, world._tree, pose, n_kp, sigma, rng)
    return RaggedFrames(xyz, label, off, pose, gt.astype(np.int64))
