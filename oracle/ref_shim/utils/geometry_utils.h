// Stand-in for the reference's utils/geometry_utils.h (PCL + Sophus): nothing of it is used by the factor sources.
#pragma once
