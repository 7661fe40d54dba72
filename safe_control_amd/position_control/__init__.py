from .cbf_qp import CBFQP, BatchedCBFQP  # noqa: F401
from .mpc_cbf import MPCCBF, BatchedMPCCBF  # noqa: F401
from .optimal_decay_mpc_cbf import OptimalDecayMPCCBF, BatchedOptimalDecayMPCCBF  # noqa: F401
from .optimal_decay_cbf_qp import OptimalDecayCBFQP, BatchedOptimalDecayCBFQP  # noqa: F401
from .manipulator_cbf_qp import ManipulatorCBFQP, BatchedManipulatorCBFQP, BatchedManipulatorTracking  # noqa: F401
from .mpc_cbf_linear import LinearMPCCBF, BatchedLinearMPCCBF, BatchedOptimalDecayLinearMPCCBF, OptimalDecayLinearMPCCBF  # noqa: F401
from .mpc_cbf_gn import GnMPCCBF, BatchedGnMPCCBF  # noqa: F401
from .mpc_cbf_vtol import VtolMPCCBF, BatchedVtolMPCCBF, OptimalDecayVtolMPCCBF, BatchedOptimalDecayVtolMPCCBF  # noqa: F401
from .backup_cbf_qp import BackupCBF, BatchedBackupCBF  # noqa: F401
from .optimal_decay_mpc_cbf_gn import OptimalDecayGnMPCCBF, BatchedOptimalDecayGnMPCCBF  # noqa: F401
