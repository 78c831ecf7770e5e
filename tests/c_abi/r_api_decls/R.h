/* compile-check declarations only: see Rinternals.h in this directory */
#include "Rinternals.h"
