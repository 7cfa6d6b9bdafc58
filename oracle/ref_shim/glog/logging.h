#include "../utils/common_ros.h"
