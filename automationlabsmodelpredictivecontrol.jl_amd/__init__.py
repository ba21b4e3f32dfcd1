"""MI355X-native batched condensed-QP MPC solve engine behind the reference's controller API.

The directory name contains a dot (the graft naming convention), so it cannot be imported with a plain
`import` statement: use `almpc_loader.load_package()` at the repository root, which registers it as
`almpc_amd`."""
from . import _capi  # noqa: F401
from . import sharding  # noqa: F401
from .controller import *  # noqa: F401,F403
from .controller import _design_reference_mpc, _model_predictive_control_design, _create_weights_coefficients  # noqa: F401
from .controller import _model_predictive_control_computation  # noqa: F401
