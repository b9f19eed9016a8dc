#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own data files.

Runs ONLY in the build container (needs /root/reference and /opt/conda/bin/h5dump).
The outputs (*.npz) are data -- inputs and expected outputs the reference's notebooks
saved -- and are committed; this script is committed next to them so they can be
regenerated.  No reference source text is copied.

Sources (all relative to /root/reference):
  savefiles/qv_kin40k.jld, Xu_kin40k.jld, params_optimal_kin40k.jld, SMSE_kin40k.jld,
  qw_kin40k.jld                       (written by experiments/regression_kin40k.ipynb:329-334)
  savefiles/qv_banana.jld, Xu_banana.jld, params_optimal_banana.jld, number_error_banana.jld,
  error_rate_banana.jld, qw_banana.jld (written by experiments/classification_banana.ipynb:347-353)
  savefiles/*_toyregression.jld, *_toyclassification.jld (experiments/GPT_regression.ipynb:2796-2800)
  savefiles/params_opt_pendulum.jld, FE_*.jld
  data/kin40k/*.mat, data/banana/banana.csv
"""
import os
import re
import subprocess
import tempfile

import numpy as np
import scipy.io

REF = "/root/reference"
SAVE = os.path.join(REF, "savefiles")
H5DUMP = "/opt/conda/bin/h5dump"
OUT = os.path.dirname(os.path.abspath(__file__))


def h5_f64(path, dataset):
    """Dump one float64 dataset as little-endian binary and return it flat."""
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run([H5DUMP, "-d", dataset, "-b", "LE", "-o", tmp.name, path],
                       check=True, stdout=subprocess.DEVNULL)
        return np.fromfile(tmp.name, dtype="<f8").copy()


def h5_text(path, dataset):
    return subprocess.run([H5DUMP, "-m", "%.17g", "-d", dataset, path], check=True,
                          capture_output=True, text=True).stdout


def h5_ref_order(path, dataset):
    """Return the /_refs/NNNNNNNN names an array-of-references dataset points to, in order."""
    txt = h5_text(path, dataset)
    return re.findall(r"(/_refs/\d{8})", txt)


def vec_of_vec(path, dataset):
    refs = h5_ref_order(path, dataset)
    return np.stack([h5_f64(path, r) for r in refs])


def compound_scalars(path, dataset):
    txt = h5_text(path, dataset)
    body = txt[txt.index("DATA {"):]
    return [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body.split("(0):", 1)[1])]


def steprange(path, dataset):
    """Julia StepRangeLen{Float64,TwicePrecision,TwicePrecision}: ref(hi,lo), step(hi,lo), len, offset."""
    v = compound_scalars(path, dataset)
    ref_hi, ref_lo, st_hi, st_lo, length, offset = v[:6]
    i = np.arange(1, int(length) + 1)
    return (ref_hi + ref_lo) + (i - int(offset)) * (st_hi + st_lo)


def sigma_summary(S):
    """Keep a 2.9 MB covariance out of the repo: diagonal, Frobenius norm, trace, first 4 rows."""
    return dict(diag=np.diag(S).copy(), fro=np.linalg.norm(S), trace=np.trace(S), rows=S[:4].copy(),
                asym=np.abs(S - S.T).max())


def main():
    # ---------------- kin40k ----------------
    d = os.path.join(REF, "data/kin40k")
    xtrain = scipy.io.loadmat(os.path.join(d, "kin40k_xtrain.mat"))["xtrain"]
    ytrain = scipy.io.loadmat(os.path.join(d, "kin40k_ytrain.mat"))["ytrain"].ravel()
    xtest = scipy.io.loadmat(os.path.join(d, "kin40k_xtest.mat"))["xtest"]
    ytest = scipy.io.loadmat(os.path.join(d, "kin40k_ytest.mat"))["ytest"].ravel()
    np.savez_compressed(os.path.join(OUT, "kin40k_data.npz"),
                        xtrain=xtrain, ytrain=ytrain, xtest=xtest, ytest=ytest)

    mu = h5_f64(os.path.join(SAVE, "qv_kin40k.jld"), "/_refs/00000001")
    Sig = h5_f64(os.path.join(SAVE, "qv_kin40k.jld"), "/_refs/00000002").reshape(600, 600)
    Xu = vec_of_vec(os.path.join(SAVE, "Xu_kin40k.jld"), "/Xu")
    theta = h5_f64(os.path.join(SAVE, "params_optimal_kin40k.jld"), "/params_optimal")
    smse = h5_f64(os.path.join(SAVE, "SMSE_kin40k.jld"), "/SMSE")
    qw = compound_scalars(os.path.join(SAVE, "qw_kin40k.jld"), "/qw")
    s = sigma_summary(Sig)
    np.savez_compressed(os.path.join(OUT, "kin40k_fixture.npz"),
                        mu_v=mu, Xu=Xu, theta_opt=theta, smse=smse, qw_ab=np.array(qw[:2]),
                        Sigma_diag=s["diag"], Sigma_fro=s["fro"], Sigma_trace=s["trace"],
                        Sigma_rows=s["rows"], Sigma_asym=s["asym"])

    # ---------------- banana ----------------
    csv = np.genfromtxt(os.path.join(REF, "data/banana/banana.csv"), delimiter=",", skip_header=1)
    mu = h5_f64(os.path.join(SAVE, "qv_banana.jld"), "/_refs/00000001")
    Sig = h5_f64(os.path.join(SAVE, "qv_banana.jld"), "/_refs/00000002").reshape(500, 500)
    Xu = vec_of_vec(os.path.join(SAVE, "Xu_banana.jld"), "/Xu")
    theta = h5_f64(os.path.join(SAVE, "params_optimal_banana.jld"), "/params_optimal")
    s = sigma_summary(Sig)
    qw = compound_scalars(os.path.join(SAVE, "qw_banana.jld"), "/qw")
    extra = {}
    for name, ds in (("number_error_banana", "/number_error"), ("error_rate_banana", "/error_rate")):
        txt = h5_text(os.path.join(SAVE, name + ".jld"), ds)
        extra[name] = float(re.search(r"\(0\):\s*([-+\d.eE]+)", txt).group(1))
    np.savez_compressed(os.path.join(OUT, "banana_fixture.npz"),
                        data=csv, mu_v=mu, Xu=Xu, theta_opt=theta, qw_ab=np.array(qw[:2]),
                        Sigma_diag=s["diag"], Sigma_fro=s["fro"], Sigma_trace=s["trace"],
                        Sigma_rows=s["rows"], Sigma_asym=s["asym"],
                        number_error=extra["number_error_banana"], error_rate=extra["error_rate_banana"])

    # ---------------- toy sets ----------------
    for kind in ("toyregression", "toyclassification"):
        arrs = {}
        for v in ("xtrain", "ytrain", "xtest", "ytest"):
            p = os.path.join(SAVE, f"{v}_{kind}.jld")
            hdr = subprocess.run([H5DUMP, "-H", p], check=True, capture_output=True, text=True).stdout
            if f'DATASET "{v}"' in hdr and "H5T_IEEE_F64LE" in hdr.split(f'DATASET "{v}"')[1][:120]:
                arrs[v] = h5_f64(p, "/" + v)
            else:  # stored as a range object
                arrs[v] = steprange(p, "/" + v)
        arrs["Xu"] = steprange(os.path.join(SAVE, f"Xu_{kind}.jld"), "/Xu")
        np.savez_compressed(os.path.join(OUT, f"{kind}_fixture.npz"), **arrs)

    # ---------------- small traces / params ----------------
    misc = {"params_opt_pendulum": h5_f64(os.path.join(SAVE, "params_opt_pendulum.jld"), "/params")}
    for name in ("FE_banana", "FE_kin40k", "FE_pendulum"):
        p = os.path.join(SAVE, name + ".jld")
        names = subprocess.run([H5DUMP, "-n", p], check=True, capture_output=True, text=True).stdout
        refs = sorted(set(re.findall(r"(/_refs/\d{8})", names)))
        if refs:
            misc[name] = np.concatenate([h5_f64(p, r) for r in refs])
    np.savez_compressed(os.path.join(OUT, "misc_fixture.npz"), **misc)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            z = np.load(os.path.join(OUT, f))
            print(f, {k: z[k].shape for k in z.files})


if __name__ == "__main__":
    main()
