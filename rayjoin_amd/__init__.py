"""rayjoin_amd -- MI355X-native LSI / PIP query path (software LBVH, hand-written HIP for gfx950)
behind RayJoin's LSI/PIP operator interface.  See DESIGN.md."""
from .maps import (Context, PlanarGraph, ScaledMap, Scaling, load_from, read_cdb,  # noqa: F401
                   write_cdb, serialize_bin, deserialize_bin, MISS, EXTERIOR_FACE_ID)

__version__ = "0.1.0"
