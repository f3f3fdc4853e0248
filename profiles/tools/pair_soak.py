#!/usr/bin/env python3
"""Soak: N fused two-network steps of train_deep_fake's denoise mode (2 x 8 x 256 x 256, fresh noise every step) against the
sequential loop on the same kernel choices, same seeds: the two runs must stay BIT-identical (losses, both parameter
buffers) for the whole run, and the losses must fall.   python3 profiles/tools/pair_soak.py [steps=200] [dtype=f32]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from denoising_diffusion_deep_fake_amd.dataset import synthetic_face_crops  # noqa: E402
from denoising_diffusion_deep_fake_amd.train_deep_fake.lit_module import LitModule  # noqa: E402
from denoising_diffusion_deep_fake_amd.trainer import optimizer_steps  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"


def run(**kw):
    torch.manual_seed(3)
    lit = LitModule(mode="denoise", batch_size=8, learning_rate=0.01, adam_b1=0.5, adam_b2=0.999, max_epochs=1,
                    cosine_scheduler_max_epoch=50, num_workers=0, encoder_name="resnet34", noise_exponential_sampling_lambda=3,
                    mean_a=[0.5] * 3, std_a=[0.5] * 3, mean_b=[0.5] * 3, std_b=[0.5] * 3, synthetic=True, image_size=256,
                    precision=dtype, augment=True, **kw).cuda().train()
    opts, _ = lit.configure_optimizers()
    opt_params = [[p for g in o.param_groups for p in g["params"]] for o in opts]
    batches = [{k: {"image": synthetic_face_crops(8, 256, seed=7 + 10 * j + i, device="cuda"), "index": None}
                for i, k in enumerate("ab")} for j in range(4)]
    torch.manual_seed(50)
    losses = []
    for it in range(steps):
        optimizer_steps(lit, opts, opt_params, batches[it % 4], it, True, None)
        if it % 10 == 9 or it == 0:
            losses.append((float(lit._logged["loss_denoise/train_a"]), float(lit._logged["loss_denoise/train_b"])))
    torch.cuda.synchronize()
    return losses, lit.model_a.flat_params.clone(), lit.model_b.flat_params.clone(), lit._pair is not None


f, s = run(), run(pair_fused=False, pair_plan=True)
out = {"steps": steps, "dtype": dtype, "fused_route": f[3] and not s[3],
       "losses_bit_identical": f[0] == s[0], "params_a_bit_identical": bool(torch.equal(f[1], s[1])),
       "params_b_bit_identical": bool(torch.equal(f[2], s[2])),
       "loss_first": f[0][0], "loss_last": f[0][-1], "finite": bool(torch.isfinite(f[1]).all() and torch.isfinite(f[2]).all())}
print(json.dumps(out))
assert out["fused_route"] and out["losses_bit_identical"] and out["params_a_bit_identical"] and out["params_b_bit_identical"]
assert out["finite"] and out["loss_last"][0] < out["loss_first"][0] and out["loss_last"][1] < out["loss_first"][1]
