"""hilo_mpc_amd - MI355X-native batched backend for HILO-MPC's solve path.

Host classes mirror the reference API surface for the hot path only:
`NMPC.setup()/optimize()`, `LMPC`, `MHE.estimate()`, `KF/EKF/UKF.estimate()/predict()/update()`,
`GaussianProcess.setup()/predict()`, `Kernel`, `Mean` - all with a leading batch axis - and the rows marked "next" there
(models written as expressions, `fit_model`, `SMPC`, `SimpleControlLoop`, `ParticleFilter`), on top of the C ABI of
`libhilo_hip.so` (include/hilo_hip.h).  There is no CPU fallback.
"""
from .model import Model
from .estimator import KalmanFilter, ExtendedKalmanFilter, UnscentedKalmanFilter
from .gp import GaussianProcess, Kernel, Mean
# the kernel and mean classes under the names of the reference's flat namespace (hilo_mpc/__init__.py:66-87)
from .gp import (ConstantKernel, SquaredExponentialKernel, MaternKernel, ExponentialKernel, Matern32Kernel, Matern52Kernel,
                 RationalQuadraticKernel, PiecewisePolynomialKernel, DotProductKernel, PolynomialKernel, LinearKernel,
                 NeuralNetworkKernel, PeriodicKernel, ConstantMean, ZeroMean, OneMean, PolynomialMean, LinearMean)

KF = KalmanFilter
EKF = ExtendedKalmanFilter
UKF = UnscentedKalmanFilter
GP = GaussianProcess

__all__ = ['Model', 'KalmanFilter', 'ExtendedKalmanFilter', 'UnscentedKalmanFilter', 'KF', 'EKF', 'UKF',
           'GaussianProcess', 'GP', 'Kernel', 'Mean', 'ConstantKernel', 'SquaredExponentialKernel', 'MaternKernel',
           'ExponentialKernel', 'Matern32Kernel', 'Matern52Kernel', 'RationalQuadraticKernel', 'PiecewisePolynomialKernel',
           'DotProductKernel', 'PolynomialKernel', 'LinearKernel', 'NeuralNetworkKernel', 'PeriodicKernel', 'ConstantMean',
           'ZeroMean', 'OneMean', 'PolynomialMean', 'LinearMean']
from .nmpc import NMPC
from .smpc import SMPC
from . import expr
from .mhe import MovingHorizonEstimator, MHE
from .lmpc import LMPC
from .control_loop import SimpleControlLoop
from .pf import ParticleFilter

PF = ParticleFilter

__all__ += ['NMPC', 'SMPC', 'MovingHorizonEstimator', 'MHE', 'LMPC', 'expr', 'SimpleControlLoop', 'ParticleFilter', 'PF']
