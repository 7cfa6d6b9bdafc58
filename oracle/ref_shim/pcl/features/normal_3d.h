// nothing of pcl/features/normal_3d.h is used by the reference PointProcessor
