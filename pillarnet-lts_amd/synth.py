"""Seeded synthetic frames and weights of the shapes the benchmark configurations name
(no dataset or checkpoint is reachable offline).  numpy PCG64 so CPU oracle and GPU path see
byte-identical inputs."""
import numpy as np


def nusc_sweep(n_points=30000, seed=0, dt=0.0, shift_x=0.0, r_max=59.0, z_mean=-1.0, z_std=1.0,
               z_clip=(-5.0, 3.0)):
    """One nuScenes-like sweep [n,5] fp32 (x, y, z, intensity, dt):  r = 1 + r_max*u^2,
    phi ~ U(0, 2pi), z ~ N clipped, intensity ~ U(0,255)."""
    rng = np.random.default_rng(seed)
    u = rng.random(n_points)
    r = 1.0 + r_max * u * u
    phi = rng.random(n_points) * 2.0 * np.pi
    x = r * np.cos(phi) + shift_x
    y = r * np.sin(phi)
    z = np.clip(rng.normal(z_mean, z_std, n_points), z_clip[0], z_clip[1])
    inten = rng.random(n_points) * 255.0
    t = np.full(n_points, dt)
    return np.stack((x, y, z, inten, t), axis=1).astype(np.float32)


def nusc_frame(sweeps=10, points_per_sweep=30000, seed=0):
    """configs[1] input: 10 sweeps x 30k points, sweep k has dt = 0.05k and a 0.5k m ego shift."""
    parts = [nusc_sweep(points_per_sweep, seed=seed * 1000 + k, dt=0.05 * k, shift_x=0.5 * k)
             for k in range(sweeps)]
    return np.concatenate(parts, axis=0)


def waymo_frame(n_points=180000, seed=100):
    """configs[2] input [n,5]: x, y, z, tanh-intensity in [0,1), elongation ~ U(0,1)."""
    rng = np.random.default_rng(seed)
    u = rng.random(n_points)
    r = 1.0 + 74.0 * u * u
    phi = rng.random(n_points) * 2.0 * np.pi
    z = np.clip(rng.normal(0.5, 1.2, n_points), -2.0, 4.0)
    inten = np.tanh(rng.random(n_points) * 2.0)
    elong = rng.random(n_points)
    return np.stack((r * np.cos(phi), r * np.sin(phi), z, inten, elong), axis=1).astype(np.float32)


def nms_boxes(n=1000, seed=400, extent=54.0, cluster=0.0):
    """Box set for the NMS micro-benchmark, pcdet frame [x,y,z,dx,dy,dz,heading]; `cluster` > 0
    draws centres around n/20 seeds so that many boxes overlap."""
    rng = np.random.default_rng(seed)
    if cluster > 0:
        k = max(1, n // 20)
        seeds = rng.uniform(-extent, extent, (k, 2))
        ctr = seeds[rng.integers(0, k, n)] + rng.normal(0, cluster, (n, 2))
    else:
        ctr = rng.uniform(-extent, extent, (n, 2))
    z = rng.normal(-1.0, 0.5, (n, 1))
    dims = np.exp(rng.normal(np.log([4.5, 1.9, 1.6]), 0.25, (n, 3)))
    th = rng.uniform(-np.pi, np.pi, (n, 1))
    return np.concatenate((ctr, z, dims, th), axis=1).astype(np.float32)


def randomize_bn_(module, seed=0):
    """Eval-mode BN must be non-trivial: mean ~ N(0,.1), var ~ U(.5,1.5), gamma ~ U(.8,1.2),
    beta ~ N(0,.1)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.4 + 0.8)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return module
