"""MI355X-native hot path of the path-space PDE solver.

The directory can be used in two ways:
  * as a package:      ``from path_space_pde_solver_amd import Solver, LLGC``
    (``path_space_pde_solver_amd/`` at the repo root is a thin import shim because this
    directory's name is not a valid Python identifier);
  * flat, like the reference: put this directory on ``sys.path`` and
    ``from solver import Solver``; ``from problems import LLGC`` -- the module names the
    reference's notebooks import.
"""
from .function_space import (Affine, Constant, DenseNet, DenseNet_tanh, DenseNet_tanh_2, Linear, MySequential,  # noqa: F401
                             SingleParam)
from .problems import (LLGC, LQGC, AllenCahn, DoubleWell, DoubleWell_multidim,  # noqa: F401
                       DoubleWell_multidim_for_general_solver, HeatEquation, ExponentialOnSphere,
                       ExponentialOnBallNonlinear, ExponentialOnBallNonlinearSin,
                       ExponentialOnSphereNonlinearParabolic, QuadraticOnBox, Committor)
from .solver import Solver  # noqa: F401
from .general_solver import GeneralSolver, EllipticSolver  # noqa: F401
from .plan_native import PlanUnsupported  # noqa: F401
from . import native  # noqa: F401
from . import native_shapes  # noqa: F401
from . import plan_native, plan_dense_native, plan_general_native, plan_value_native  # noqa: F401
from .utilities import do_importance_sampling_me  # noqa: F401

__all__ = ['Solver', 'GeneralSolver', 'EllipticSolver', 'ExponentialOnSphere', 'ExponentialOnBallNonlinear',
           'ExponentialOnBallNonlinearSin', 'ExponentialOnSphereNonlinearParabolic', 'QuadraticOnBox', 'Committor', 'LLGC', 'LQGC', 'DoubleWell', 'DoubleWell_multidim', 'DoubleWell_multidim_for_general_solver',
           'AllenCahn', 'HeatEquation', 'MySequential', 'DenseNet', 'DenseNet_tanh', 'DenseNet_tanh_2', 'SingleParam',
           'Constant', 'Linear', 'Affine', 'PlanUnsupported', 'native']
