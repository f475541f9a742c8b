#include "boost_double.h"
