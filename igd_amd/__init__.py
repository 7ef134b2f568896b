"""igd_amd -- MI355X-native overlap search for IGD databases (databio/IGD's `igd search` hot
path), behind the reference's own C ABI.  See DESIGN.md / INTEGRATION.md."""
from ._native import NativeMissing, build  # noqa: F401

__all__ = ["Database", "NativeMissing", "build"]
# `from igd_amd import igd_py as iGD; iGD.igd_py()` mirrors the reference's `import igd_py as iGD`


def __getattr__(name):
    if name == "Database":
        from .database import Database
        return Database
    raise AttributeError(name)
