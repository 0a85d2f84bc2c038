"""dapol_amd: MI355X-native DAPOL+ proving path (libdapol_hip.so + a thin ctypes binding)."""
from . import capi  # noqa: F401
