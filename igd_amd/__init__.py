"""igd_amd -- MI355X-native overlap search for IGD databases (databio/IGD's `igd search` hot
path), behind the reference's own C ABI.  See DESIGN.md / INTEGRATION.md."""
from ._native import NativeMissing, build  # noqa: F401

__all__ = ["Database", "igd_py", "NativeMissing", "build"]


def __getattr__(name):
    if name == "Database":
        from .database import Database
        return Database
    if name == "igd_py":
        from .igd_py import igd_py
        return igd_py
    raise AttributeError(name)
