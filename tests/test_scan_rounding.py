"""Why the forward kernels scan a chunk's transmittance in two forms (relu_field_kernels.hip: wave_incl_scan_trans): a float32 numpy
re-enactment of the three scan orders on the ray that showed the problem in the randomised sweep -- 1000 samples of a slowly varying
low density (alpha ~ 1.25e-4) and a last sample that takes what is left (alpha = 1: the reference's 1e10-long last interval inside
the volume), so that the accumulated weight is exactly 1 in exact arithmetic.

  * torch.cumprod's sequential product (the reference): each rounding is independent of the last -- no drift;
  * a doubling (Hillis-Steele / DPP) PRODUCT scan per 64-sample chunk: neighbouring factors are the same float, the first doubling
    step rounds every lane the same way and the later steps multiply that error by 32 per chunk -- 1e-5 of transmittance per 1000 samples;
  * the same scan carrying a = 1 - prod E beside e = prod E (a' = a_prefix + a e_prefix): the roundings are relative to a ~ 1e-2.

No GPU, no library: pure host arithmetic (the GPU-side evidence is tests/test_hip_parity_fuzz.py::test_rays_of_255_to_5000_samples)."""
import numpy as np

f32 = np.float32


def _fog_ray(S=1000, seed=0):
    rng = np.random.default_rng(seed)
    x = (1.25e-4 * (1.0 + 0.03 * np.sin(np.arange(S) / 40.0) + 0.002 * rng.standard_normal(S))).astype(np.float32)  # sigma * delta
    E = np.exp(-x.astype(np.float64)).astype(np.float32)  # correctly rounded float32 exponential
    E[-1] = f32(0.0)  # the last interval is 1e10 long
    return (f32(1.0) - E).astype(np.float32), E  # alpha = 1 - E exactly (Sterbenz), as in the reference


def _weights_sum(alpha, T):
    return float(np.sum((alpha * T).astype(np.float32).astype(np.float64)))


def _sequential(E):
    T = np.empty_like(E)
    c = f32(1.0)
    for i in range(E.size):
        T[i] = c
        c = f32(c * E[i])
    return T


def _chunk_scan(alpha, E, two_forms: bool):
    S = E.size
    T = np.empty_like(E)
    carry = f32(1.0)
    for c0 in range(0, S, 64):
        m = min(64, S - c0)
        e = np.ones(64, np.float32)
        a = np.zeros(64, np.float32)
        e[:m], a[:m] = E[c0 : c0 + m], alpha[c0 : c0 + m]
        sh = 1
        while sh < 64:  # inclusive doubling scan: lane i joins the prefix that ends at lane i - sh
            ep, ap = e.copy(), a.copy()
            ep[sh:], ap[sh:] = e[:-sh], a[:-sh]
            ep[:sh], ap[:sh] = f32(1.0), f32(0.0)
            a = (a.astype(np.float64) * ep.astype(np.float64) + ap.astype(np.float64)).astype(np.float32)  # one rounding: fmaf(a, e_prefix, a_prefix)
            e = (e * ep).astype(np.float32)
            sh *= 2
        incl = np.where(a < f32(0.25), f32(1.0) - a, e).astype(np.float32) if two_forms else e
        excl = np.concatenate([[f32(1.0)], incl[:-1]]).astype(np.float32)
        T[c0 : c0 + m] = (carry * excl[:m]).astype(np.float32)
        carry = f32(carry * incl[63])
    return T


def test_a_product_scan_drifts_on_slowly_varying_density_and_the_two_form_scan_does_not():
    worst = {"sequential": 0.0, "product": 0.0, "two_forms": 0.0}
    for seed in range(4):
        alpha, E = _fog_ray(seed=seed)
        worst["sequential"] = max(worst["sequential"], abs(_weights_sum(alpha, _sequential(E)) - 1.0))
        worst["product"] = max(worst["product"], abs(_weights_sum(alpha, _chunk_scan(alpha, E, False)) - 1.0))
        worst["two_forms"] = max(worst["two_forms"], abs(_weights_sum(alpha, _chunk_scan(alpha, E, True)) - 1.0))
    assert worst["sequential"] < 2e-6, worst  # the reference's order: the weights telescope (independent roundings)
    assert worst["product"] > 3e-6, worst  # coherent rounding of the doubling steps: beyond the 1e-5 bar at a few thousand samples
    assert worst["two_forms"] < 1e-6, worst  # what the kernels do


def test_exact_zeros_behind_an_opaque_sample_survive_the_two_form_scan():
    alpha = np.full(200, f32(1e-3), np.float32)
    alpha[70] = f32(1.0)
    E = (f32(1.0) - alpha).astype(np.float32)
    T = _chunk_scan(alpha, E, True)
    assert np.all(T[71:] == 0.0) and np.all(T[:71] > 0.0)
    np.testing.assert_allclose(T[:71], _sequential(E)[:71], rtol=2e-6)
