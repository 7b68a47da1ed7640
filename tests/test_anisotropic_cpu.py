"""The anisotropic half of the textureSampleGrad contract (AWSM_CFG_ANISOTROPIC; oracle_shade.c: sample_array_grad, kernels_shade.hip: grad_footprint):
properties of the oracle's restatement.  WebGPU leaves anisotropy to the implementation (the reference asks for max_anisotropy 16 on its glTF samplers,
gltf/populate/material.rs:892-902, and samples with textureSampleGrad, texture_uvs.wgsl:122), so there is no golden vector to pin it with — what is
checked is that the rule reduces to the isotropic one, is continuous where the probe count changes, conserves energy, and does what anisotropic
filtering is for."""
import numpy as np
import pytest

from oracle import oracle_lib

LINEAR16 = {"address_mode_u": 1, "address_mode_v": 1, "mag_filter": 1, "min_filter": 1, "mipmap_filter": 1, "max_anisotropy": 16}


def _stripes(size=256, period=2):
    """vertical stripes: the value changes along u only"""
    t = np.zeros((1, size, size, 4), dtype=np.uint8)
    t[0, :, (np.arange(size) // (period // 2)) % 2 == 0, :3] = 255
    t[..., 3] = 255
    return t


def _noise(size=128, seed=3):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, size=(1, size, size, 4), dtype=np.uint8)
    return t


def test_isotropic_footprints_and_max_anisotropy_1_are_the_isotropic_rule():
    tex = _noise()
    rng = np.random.default_rng(0)
    n = 500
    uv = rng.random((n, 2)).astype(np.float32) * 3 - 1
    ang = rng.random(n) * 6.28
    r = (rng.random(n) * 8 + 0.2) / 128
    ddx = np.stack([np.cos(ang) * r, np.sin(ang) * r], axis=1).astype(np.float32)
    ddy = np.stack([-np.sin(ang) * r, np.cos(ang) * r], axis=1).astype(np.float32)      # same length, perpendicular: N = 1
    iso = oracle_lib.sample_grad(tex, LINEAR16, uv, ddx, ddy, anisotropic=False)
    ani = oracle_lib.sample_grad(tex, LINEAR16, uv, ddx, ddy, anisotropic=True)
    assert np.abs(iso - ani).max() <= 2e-6          # rho_x == rho_y up to rounding: N within an ulp of 1
    # a stretched footprint, but the sampler asks for none / has a nearest filter: unchanged
    ddy4 = ddy * 0.25
    iso = oracle_lib.sample_grad(tex, LINEAR16, uv, ddx, ddy4, anisotropic=False)
    for smp in (dict(LINEAR16, max_anisotropy=1), dict(LINEAR16, mipmap_filter=0), dict(LINEAR16, min_filter=0)):
        ref = oracle_lib.sample_grad(tex, smp, uv, ddx, ddy4, anisotropic=False)
        assert (oracle_lib.sample_grad(tex, smp, uv, ddx, ddy4, anisotropic=True) == ref).all()
    assert np.abs(oracle_lib.sample_grad(tex, LINEAR16, uv, ddx, ddy4, anisotropic=True) - iso).max() > 1e-3       # ... and with 16 it is not


def test_continuous_in_the_anisotropy_ratio():
    """The probe count changes at odd ratios (m = ceil((N - 1) / 2)) and N saturates at max_anisotropy: no jump anywhere."""
    tex = _noise(seed=5)
    ratios = np.linspace(0.8, 20.0, 4000).astype(np.float32)
    uv = np.tile(np.array([[0.37, 0.61]], dtype=np.float32), (ratios.size, 1))
    minor = 1.5 / 128
    ddx = np.stack([ratios * minor, np.zeros_like(ratios)], axis=1)
    ddy = np.tile(np.array([[0.0, minor]], dtype=np.float32), (ratios.size, 1))
    c = oracle_lib.sample_grad(tex, LINEAR16, uv, ddx, ddy, anisotropic=True)
    step = np.abs(np.diff(c, axis=0)).max(axis=1)
    assert step.max() < 4e-3, (float(step.max()), float(ratios[step.argmax()]))      # 0.5 % of the ratio per step on white noise
    # the major axis is whichever derivative is longer: the same footprints with ddx and ddy exchanged give the same colours
    c2 = oracle_lib.sample_grad(tex, LINEAR16, uv, ddy, ddx, anisotropic=True)
    assert np.abs(c2 - c).max() <= 1e-6
    # ... and a transposed texture sampled with transposed coordinates and derivatives is the same image
    tex_t = np.ascontiguousarray(tex.transpose(0, 2, 1, 3))
    sw = lambda a: np.ascontiguousarray(a[:, ::-1])
    c3 = oracle_lib.sample_grad(tex_t, LINEAR16, sw(uv), sw(ddx), sw(ddy), anisotropic=True)
    assert np.abs(c3 - c).max() <= 1e-6


def test_constant_texture_stays_constant_and_weights_are_normalised():
    tex = np.full((1, 64, 64, 4), 137, dtype=np.uint8)
    rng = np.random.default_rng(1)
    n = 300
    uv = rng.random((n, 2)).astype(np.float32)
    ddx = (rng.standard_normal((n, 2)) * 0.05).astype(np.float32)
    ddy = (rng.standard_normal((n, 2)) * 0.004).astype(np.float32)
    c = oracle_lib.sample_grad(tex, LINEAR16, uv, ddx, ddy, anisotropic=True)
    assert np.abs(c - 137.0 / 255.0).max() < 1e-6


def test_detail_across_the_minor_axis_survives():
    """A floor seen at a grazing angle: stripes along the view direction, the footprint 12 texels long and 0.75 wide.  The isotropic rule picks the level
    for 12 texels and returns grey; with max_anisotropy 16 the level is picked for the width and the stripes stay."""
    tex = _stripes(256, 4)                     # two texels white, two black, along u
    n = 64
    u = (np.arange(n) + 0.5) / n * (8 / 256)   # across two periods
    uv = np.stack([u, np.full(n, 0.3)], axis=1).astype(np.float32)
    ddx = np.tile(np.array([[0.75 / 256, 0.0]], dtype=np.float32), (n, 1))     # across the stripes: narrow
    ddy = np.tile(np.array([[0.0, 12.0 / 256]], dtype=np.float32), (n, 1))     # along the stripes: long
    iso = oracle_lib.sample_grad(tex, LINEAR16, uv, ddx, ddy, anisotropic=False)[:, 0]
    ani = oracle_lib.sample_grad(tex, LINEAR16, uv, ddx, ddy, anisotropic=True)[:, 0]
    assert iso.max() - iso.min() < 0.05 and abs(iso.mean() - 0.5) < 0.02
    assert ani.max() - ani.min() > 0.8 and abs(ani.mean() - 0.5) < 0.05
    # max_anisotropy 4: N saturates at 4, the level is chosen for 3 texels -> in between
    mid = oracle_lib.sample_grad(tex, dict(LINEAR16, max_anisotropy=4), uv, ddx, ddy, anisotropic=True)[:, 0]
    assert 0.05 < mid.max() - mid.min() < ani.max() - ani.min()
