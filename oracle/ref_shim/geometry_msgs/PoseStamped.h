#pragma once
#include "../utils/common_ros.h"
