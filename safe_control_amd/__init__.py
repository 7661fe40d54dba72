"""MI355X-native batched CBF-QP / MPC-CBF solve engine (drop-in for the
position_control.cbf_qp / mpc_cbf plugins of tkkim-robot/safe_control).

The compute path is hand-written HIP for gfx950 behind a C-ABI
(include/safe_control_amd.h, safe_control_amd/csrc).  This Python package is
the host-side mirror of the reference's plugin interface; it has no CPU
fallback and raises if the HIP library is missing.
"""
from ._lib import HipLibraryError, load as load_library  # noqa: F401
from .position_control.cbf_qp import CBFQP, BatchedCBFQP  # noqa: F401
from .position_control.manipulator_cbf_qp import ManipulatorCBFQP, BatchedManipulatorCBFQP, BatchedManipulatorTracking  # noqa: F401
from .position_control.mpc_cbf import MPCCBF, BatchedMPCCBF  # noqa: F401
from .position_control.mpc_cbf_linear import LinearMPCCBF, BatchedLinearMPCCBF, BatchedOptimalDecayLinearMPCCBF, OptimalDecayLinearMPCCBF  # noqa: F401
from .position_control.mpc_cbf_gn import GnMPCCBF, BatchedGnMPCCBF  # noqa: F401
from .position_control.mpc_cbf_vtol import VtolMPCCBF, BatchedVtolMPCCBF, OptimalDecayVtolMPCCBF, BatchedOptimalDecayVtolMPCCBF  # noqa: F401
from .position_control.mpc_cbf_ms import BatchedMSMPCCBF  # noqa: F401
from .position_control.mpc_cbf_vtol_ms import BatchedVtolMSMPCCBF, BatchedOptimalDecayVtolMSMPCCBF  # noqa: F401
from .position_control.backup_cbf_qp import BackupCBF, BatchedBackupCBF  # noqa: F401
from .position_control.optimal_decay_cbf_qp import OptimalDecayCBFQP, BatchedOptimalDecayCBFQP  # noqa: F401
from .position_control.optimal_decay_mpc_cbf import OptimalDecayMPCCBF, BatchedOptimalDecayMPCCBF  # noqa: F401
from .position_control.optimal_decay_mpc_cbf_gn import OptimalDecayGnMPCCBF, BatchedOptimalDecayGnMPCCBF  # noqa: F401
from .robots.spec import RobotHandle, complete_robot_spec  # noqa: F401
from .tracking import BatchedTrackingController  # noqa: F401

__version__ = "0.1.0"
