"""zk-mpc_amd: MI355X-native Groth16 proving hot path of zk-mpc (BLS12-377), host-side mirror.

Import as `zk_mpc_amd` (the directory name carries a hyphen; the repo-root shim package
`zk_mpc_amd/` forwards here).  Everything computes in libzkmpc_hip.so (HIP, gfx950); there is
no CPU implementation in this package.
"""
from ._lib import ZkError, load, LIB_PATH  # noqa: F401
from .api import Context, DevBuf, Bases, R1cs, ProvingKey  # noqa: F401
from . import convert  # noqa: F401
