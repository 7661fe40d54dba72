from .cbf_qp import CBFQP, BatchedCBFQP  # noqa: F401
