// stand-in for <opencv2/opencv.hpp>: see cvstub_core.h (compile check of adapter/*.cc only)
#include "../cvstub_core.h"
