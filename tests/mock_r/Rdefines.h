/* tests/mock_r/Rdefines.h -- NOT R: see Rinternals.h in this directory */
#include "Rinternals.h"
#define MAKE_CLASS(name) mock_make_class(name)
#define NEW_OBJECT(klass) mock_new_object(klass)
#define SET_SLOT(obj, name, value) mock_set_slot(obj, name, value)
#define GET_SLOT(obj, name) mock_get_slot(obj, name)
