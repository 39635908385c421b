// stand-in for <opencv/cv.h>: see cvstub_core.h (compile check of adapter/*.cc only)
#include "../cvstub_core.h"
