// pcl::CropBox is only mentioned in comments of the reference MapBuilder
