#include "point_types.h"
