#include "cv_double.h"
