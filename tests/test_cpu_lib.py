"""CPU: the C-ABI library loads, exports every symbol include/d3f_hip.h declares, and its host-side
planning (no compute, no GPU) agrees with the module tree and the survey's analytic figures."""
import ctypes as C

import pytest
import torch

from denoising_diffusion_deep_fake_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    declared = _lib.header_symbols()
    assert len(declared) >= 40
    assert set(declared) == set(_lib.PROTOTYPES), set(declared) ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.d3f_version() >= 100


def test_plan_matches_survey_figures():
    lib = _lib.lib()
    h = C.c_void_p()
    _lib.check(lib.d3f_unet_create(b"resnet34", 3, 3, 16, 256, 256, _lib.F32, C.byref(h)))
    try:
        assert lib.d3f_unet_param_floats(h) == 24_436_659
        assert lib.d3f_unet_num_params(h) == 140 and lib.d3f_unet_num_bn(h) == 46
        assert lib.d3f_unet_bnstat_floats(h) == 2 * 9504
        # SURVEY.md 8d / BASELINE.md: 15.666 GFLOP forward, 46.689 GFLOP per trained image at 256x256
        fwd = lib.d3f_unet_forward_flops(h) / 16 / 1e9
        bwd = lib.d3f_unet_backward_flops(h) / 16 / 1e9
        assert abs(fwd - 15.666) < 1e-3 and abs(fwd + bwd - 46.689) < 2e-3
        assert lib.d3f_unet_num_segments(h) == 4
        prev_begin = None
        total = 0
        for s in range(4):
            b, e = C.c_int64(), C.c_int64()
            _lib.check(lib.d3f_unet_segment_range(h, s, C.byref(b), C.byref(e)))
            assert b.value < e.value
            if prev_begin is not None:
                assert e.value == prev_begin  # buckets tile the flat gradient back to front
            prev_begin = b.value
            total += e.value - b.value
        assert prev_begin == 0 and total == 24_436_659
        assert lib.d3f_unet_workspace_bytes(h) > 2**30
    finally:
        lib.d3f_unet_destroy(h)


def test_plan_errors_mirror_smp():
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.d3f_unet_create(b"resnet34", 3, 3, 2, 48, 64, _lib.F32, C.byref(h)) != 0
    assert b"divisible by 32" in lib.d3f_last_error()
    assert lib.d3f_unet_create(b"vgg16", 3, 3, 2, 64, 64, _lib.F32, C.byref(h)) != 0
    assert b"Wrong encoder name" in lib.d3f_last_error()


def test_module_tree_matches_engine_table_and_oracle_keys():
    import oracle
    from denoising_diffusion_deep_fake_amd import Unet
    from denoising_diffusion_deep_fake_amd.unet import param_table
    net = Unet("resnet34", None, 3, 3, None)
    params, bns, nparam, nbn = param_table("resnet34", 3, 3)
    assert [n for n, _ in net.named_parameters()] == [n for n, _, _ in params]
    assert all(tuple(p.shape) == s for (_, p), (_, s, _) in zip(net.named_parameters(), params))
    assert all(off % 4 == 0 for _, _, off in params)  # 16-byte aligned views
    ref = oracle.Unet("resnet34", None, 3, 3, None)
    assert list(ref.state_dict().keys()) == list(net.state_dict().keys())
    mods = dict(net.named_modules())
    assert all(isinstance(mods[p], torch.nn.BatchNorm2d) and mods[p].num_features == c for p, c, _, _ in bns)
    r18 = Unet("resnet18", None, 3, 3, None)
    assert sum(p.numel() for p in r18.parameters()) == param_table("resnet18", 3, 3)[2]


def test_no_cpu_fallback():
    from denoising_diffusion_deep_fake_amd import D3FError, Unet, ops
    with pytest.raises(D3FError):
        Unet("resnet34", None, 3, 3, None)(torch.zeros(1, 3, 32, 32))
    with pytest.raises(D3FError):
        ops.mse_ssim_loss(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))


def test_conv_descriptor_validation_on_host():
    lib = _lib.lib()
    d = _lib.ConvDesc(2, 16, 16, 6, 0, 0, 8, 3, 3, 1, 1, 6)  # C0 = 6 not a multiple of 4
    assert lib.d3f_conv_packed_bytes(_lib.F32, C.byref(d), 0) == 0
    assert b"multiples of 4" in lib.d3f_last_error()
    d = _lib.ConvDesc(2, 16, 16, 64, 0, 0, 128, 3, 3, 2, 1, 64)
    assert lib.d3f_conv_packed_bytes(_lib.F32, C.byref(d), 0) == 128 * 576 * 4
    tiles = C.c_int()
    assert lib.d3f_conv_stats_floats(_lib.F32, C.byref(d), 0, C.byref(tiles)) == tiles.value * 128 * 2
    assert lib.d3f_conv_workspace_bytes(_lib.F32, C.byref(d), 0) > 0  # 128 rows x 128 channels: split-K


def test_library_digest_matches_sources_and_a_stale_library_is_refused(tmp_path):
    """The loaded library carries the digest of the sources it was built from (Makefile -> d3f_source_digest); a library
    sitting next to sources it was NOT built from must not load silently."""
    import shutil
    import subprocess
    import sys
    from pathlib import Path
    assert _lib.built_digest() == _lib.source_digest()
    # a copy of the package tree with the built library and one kernel source touched afterwards
    root = Path(_lib.__file__).resolve().parent.parent
    pkg = tmp_path / "denoising_diffusion_deep_fake_amd"
    shutil.copytree(root / "denoising_diffusion_deep_fake_amd", pkg,
                    ignore=shutil.ignore_patterns("build", "__pycache__", "*.o"))
    shutil.copytree(root / "include", tmp_path / "include")
    with open(pkg / "csrc" / "optim.hip", "a") as f:
        f.write("\n// edited after the build\n")
    code = ("from denoising_diffusion_deep_fake_amd import _lib\n"
            "try:\n    _lib.lib()\nexcept _lib.D3FError as e:\n    print('REFUSED', e)\nelse:\n    print('LOADED')\n")
    env = {k: v for k, v in __import__("os").environ.items() if k != "D3F_LIB"}
    out = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert "REFUSED" in out.stdout and "built from other sources" in out.stdout, (out.stdout, out.stderr)
    # ... while naming the same file explicitly as a variant build loads it
    env["D3F_LIB"] = str(pkg / "csrc" / "libd3f_hip.so")
    out = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert "LOADED" in out.stdout, (out.stdout, out.stderr)


def test_plan_counts_at_the_headline_shapes():
    """d3f_unet_plan_counts: which kernel family the plan chose for every launch of a training step (planning needs no
    GPU).  A shape or dtype that silently falls back to the implicit GEMM would only show as a slower step; this pins the
    plans the round-5 numbers were measured on: fp32 16x256x256 (7 Winograd + 3 patch / stem forward launches, 5 patch
    weight gradients) and bf16 16x256x256 (32 patch-resident forward + 35 data-gradient launches, 6 + 6 patch-kernel
    launches, the bf16 stem, 22 native-bf16 patch weight gradients); other extents keep the implicit GEMM for the wide
    layers."""
    import ctypes as C
    from denoising_diffusion_deep_fake_amd import _lib
    L = _lib.lib()

    def plan(dtype, B, S):
        h = C.c_void_p()
        assert L.d3f_unet_create(b"resnet34", 3, 3, B, S, S, dtype, C.byref(h)) == 0, L.d3f_last_error()
        f, d, w = (C.c_int32 * 16)(), (C.c_int32 * 16)(), (C.c_int32 * 16)()
        assert L.d3f_unet_plan_counts(h, f, d, w) == 0
        L.d3f_unet_destroy(h)
        return list(f), list(d), list(w)

    f, d, w = plan(_lib.F32, 16, 256)
    assert sum(f) == 47 and f[15] == 7 and f[1] == 2 and f[2] == 1 and f[0] == 37       # Winograd x 7, conv_patch x 2, conv_stem
    assert sum(d) == 50 and d[1] == 2 and d[0] == 48
    assert sum(w) == 51 and w[0] == 44 and w[1] + w[2] + w[3] + w[4] == 7 and w[7] == 0  # (class + skip passes count twice)
    f, d, w = plan(_lib.BF16, 16, 256)
    assert f[9:13] == [7, 8, 12, 5] and f[8] == 1 and f[3] == 2 and f[4] == 1 and f[5] == 1 and f[0] == 10 and f[15] == 0
    assert d[9:13] == [8, 9, 13, 5] and d[3] == 1 and d[4] == 2 and d[6] == 1 and d[7] == 1 and d[0] == 10
    assert w[7] == 22 and w[6] == 1 and w[0] == 25 and sum(w) == 48
    # 128-wide input and the authors' 448: the wide layers' extents do not match the patch-resident forms -> implicit GEMM
    for B, S in ((16, 128), (2, 448)):
        f, d, w = plan(_lib.BF16, B, S)
        assert sum(f[9:13]) == 0 and sum(d[9:13]) == 0 and f[8] == 1 and w[7] == 22
