/* tests/mock_r/R.h -- NOT R: see Rinternals.h in this directory */
#include "Rinternals.h"
