import numpy as np, sys
sys.path.insert(0, '.')
import gaussianprocessnode_amd as G
from oracle import sgp_oracle as O
fx = np.load('tests/golden/kin40k_fixture.npz')
s2, ell = O.kernel_from_theta(fx['theta_opt'], True)
K = O.kernelmatrix(s2, ell, fx['Xu'])
print('cond', np.linalg.cond(K), 'min eig', np.linalg.eigvalsh(K)[0])
for trial in range(3):
    try:
        L = G.potrf(K)
        print('potrf ok', np.abs(L - np.linalg.cholesky(K)).max())
    except Exception as e:
        print('potrf fail', e)
kd = np.load('tests/golden/kin40k_data.npz')
for trial in range(3):
    with G.SGPDevice(10000, 600, 8) as dev:
        dev.set_inducing(fx['Xu']); dev.set_data(kd['xtrain'], kd['ytrain']); dev.set_kernel(s2, ell, 0.0)
        dev.set_prior_isotropic(50.0); dev.set_noise([[1e4]])
        dev.sweep()
        try:
            mu, S, U = dev.posterior(); print('sweep ok')
        except Exception as e:
            print('sweep fail', str(e)[:120])
