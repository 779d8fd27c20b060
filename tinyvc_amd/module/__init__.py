"""Drop-in mirror of the reference's `module` package for the inference path:
`module.tinyvc`, `module.infer`, `module.utils` with the reference's names and signatures,
executing on libtinyvc_hip.so."""
from . import tinyvc, infer, utils  # noqa: F401
