/* igd_base.h -- name the reference's Cython wrapper includes (src_py/igd_py.pyx:6-19); see ../igd_py_abi.h */
#include "../igd_py_abi.h"
