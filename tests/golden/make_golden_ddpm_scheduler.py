"""Golden vectors for the demo path's scheduler boundary (reference models/lion.py:24-26, 39-40, 55, 70: ``LION.sample``
drives ``diffusers.DDPMScheduler``; diffusers is pinned to 0.11.1 in the reference's env.yaml:155 and is neither vendored
in /root/reference nor installed here).  This script RESTATES the published algorithm of that version --
``DDPMScheduler.step`` / ``_get_variance`` (src/diffusers/schedulers/scheduling_ddpm.py, v0.11.1) -- in float64 numpy:

    alpha_prod_t = abar[t];  alpha_prod_t_prev = abar[t-1] (1 at t = 0);  beta_prod_t = 1 - abar[t]
    x0 = (x - sqrt(beta_prod_t) eps) / sqrt(alpha_prod_t)                                    (prediction_type 'epsilon')
    mean = sqrt(alpha_prod_t_prev) beta_t / beta_prod_t * x0 + sqrt(alpha_t) (1 - alpha_prod_t_prev) / beta_prod_t * x
    variance = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * beta_t                         (computed first)
        'fixed_small': clamp(variance, min=1e-20);  'fixed_large': beta_t;  any other string (the reference passes
        cfg.ddpm.model_var_type = 'fixedlarge', which is not one of diffusers' names): no branch matches -> unclamped
    prev = mean + sqrt(variance) z   for t > 0,   prev = mean   for t = 0
on the schedule of the released airplane config (linear betas 1e-4 .. 0.02, 1000 steps), for a few (t, x, eps, z), and
writes tests/golden/ddpm_scheduler_0_11_1.json.  Parity at this boundary stays "unpinned by the reference" (the
dependency cannot be executed here); the fixture pins lion_amd's shim to the published algorithm and against refactors."""
import json
import os

import numpy as np


def diffusers_0_11_1_step(betas, t, x, eps, z, variance_type):
    alphas = 1.0 - betas
    abar = np.cumprod(alphas)
    ap_t = abar[t]
    ap_prev = abar[t - 1] if t > 0 else 1.0
    bp_t, bp_prev = 1.0 - ap_t, 1.0 - ap_prev
    x0 = (x - np.sqrt(bp_t) * eps) / np.sqrt(ap_t)
    mean = np.sqrt(ap_prev) * betas[t] / bp_t * x0 + np.sqrt(alphas[t]) * bp_prev / bp_t * x
    var = bp_prev / bp_t * betas[t]
    if variance_type == "fixed_small":
        var = max(var, 1e-20)
    elif variance_type == "fixed_large":
        var = betas[t]
    return mean + (np.sqrt(var) * z if t > 0 else 0.0), float(var)


def main():
    betas = np.linspace(1e-4, 0.02, 1000, dtype=np.float64)
    rng = np.random.default_rng(20260924)
    cases = []
    for t in (999, 500, 37, 1, 0):
        x, eps, z = rng.standard_normal(6), rng.standard_normal(6), rng.standard_normal(6)
        for vt in ("fixedlarge", "fixed_small", "fixed_large"):
            prev, var = diffusers_0_11_1_step(betas, t, x, eps, z, vt)
            cases.append({"t": t, "variance_type": vt, "x": x.tolist(), "eps": eps.tolist(), "z": z.tolist(),
                          "prev_sample": np.asarray(prev).tolist(), "variance": var})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ddpm_scheduler_0_11_1.json")
    json.dump({"schedule": "linear 1e-4 .. 0.02, 1000 steps (config/airplane_prior_cfg.yml via default_config.py)",
               "source": "diffusers v0.11.1 scheduling_ddpm.py, restated (not executed)", "cases": cases}, open(out, "w"), indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
