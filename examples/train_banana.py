"""The reference's banana classification experiment end to end on one MI355X: `PerformInference` of
experiments/classification_banana.ipynb (N = 4000, M = 500, minibatches of 200, 500 epochs of
[Probit moment matching -> VMP sweep for q(v) -> Gamma update of q(w) -> AdaMax step on theta], posterior carried over
every minibatch without reset), then the 1300-point test prediction.

The reference reports 125 errors (rate 0.0961538) after 2965.757395 s.  Data and inducing inputs are the committed
golden fixtures (tests/golden/banana_fixture.npz).  The final theta and q(w) rate differ from the reference's saved ones
(softplus(theta) = [0.986, 1.028, 1.022], rate 1.72e6): the q(w) / theta dynamics of this model are neutrally stable and
end where the message schedule puts them (tests/scripts/banana_schedules.py measures seven).  Prints one
JSON line.  --host-paced runs the same loop through the setters instead of sgp_train_*.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianprocessnode_amd import hostbind  # noqa: E402

hostbind.bind_to_gpu_node(0)      # the host side on the GPU's NUMA node, before the HIP runtime starts


def run(epochs=500, batch=200, device_paced=True):
    import gaussianprocessnode_amd as G
    from gaussianprocessnode_amd.meta import softplus
    from gaussianprocessnode_amd.train import perform_inference_classification

    fix = np.load(os.path.join(ROOT, "tests", "golden", "banana_fixture.npz"))
    data = fix["data"]
    X, lab = data[:, :2], np.where(data[:, 2] < 0, 0.0, data[:, 2])      # float(replace(x, -1 => 0))
    Ntrain = 4000
    xtrain, ytrain, xtest, ytest = X[:Ntrain], lab[:Ntrain], X[Ntrain:], lab[Ntrain:]
    Xu = fix["Xu"]
    M, D = Xu.shape
    theta_init = np.log(np.expm1(np.ones(D + 1)))
    with G.SGPDevice(batch, M, D) as dev:
        t0 = time.perf_counter()
        qv, (a, b), theta = perform_inference_classification(theta_init, xtrain, ytrain, Xu, dev, batch_size=batch,
                                                             epochs=epochs, device_paced=device_paced)
        t_train = time.perf_counter() - t0
        p = softplus(theta)
        dev.set_kernel(float(p[0]), p[1:], 1e-8)
        pred = dev.predict(xtest, qv.m)
    errors = float(np.sum(np.abs((pred >= 0).astype(float) - ytest)))      # mean(Probit(:out)) >= 0.5  <=>  mean f >= 0
    return {
        "experiment": "banana PerformInference (experiments/classification_banana.ipynb)",
        "epochs": epochs, "minibatch": batch, "M": int(M), "pacing": "device" if device_paced else "host", "train_seconds": t_train,
        "ms_per_minibatch": 1e3 * t_train / (epochs * (Ntrain // batch)),
        "errors": errors, "error_rate": errors / len(ytest), "theta_softplus": [float(v) for v in p], "qw": [a, b],
        "reference": {"errors": 125.0, "error_rate": 0.09615384615384616, "train_seconds": 2965.757395,
                      "theta_softplus": [float(v) for v in softplus(fix["theta_opt"])], "qw": [float(v) for v in fix["qw_ab"]]},
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=500)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--host-paced", action="store_true", help="setters + sgp_theta_objective + AdaMax in NumPy per minibatch")
    args = ap.parse_args()
    print(json.dumps(run(args.epochs, args.batch, not args.host_paced)))
