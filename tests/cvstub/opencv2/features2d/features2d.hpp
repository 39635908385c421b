// stand-in for <opencv2/features2d/features2d.hpp>: see cvstub_core.h (compile check of adapter/*.cc only)
#include "../../cvstub_core.h"
