"""orb-slam2_amd — MI355X-native ORB front end + Hamming matchers for ORB-SLAM2.

Host-side mirror (Python, ctypes) of the reference's operator interface for the hot path,
sitting directly on the C ABI of liborbx.so (include/orbx.h):

  ORBextractor          <- reference include/ORBextractor.h:58-139
  ORBmatcher            <- reference include/ORBmatcher.h:41-103 (the three north-star searches)
  ComputeStereoMatches  <- reference src/Frame.cc:577-751

There is NO CPU fallback: importing works without a GPU (so that the ABI can be inspected), but
every compute call raises OrbxError unless liborbx.so is built and a gfx950 device is present.
The directory name contains a '-', so load it with importlib (see tests/conftest.py: load_pkg()).
"""
from . import orbx, streams
from .orbx import (OrbxError, KP_DTYPE, lib, lib_path, ORBextractor, ORBmatcher, ComputeStereoMatches,
                   FeatSet, make_featset, STAGES, BowDatabase, DeviceKeyFrame, BowFrames, ORBVocabulary, ComputeDistinctiveDescriptors, Rectifier, UndistortKeyPoints)

__all__ = ["OrbxError", "KP_DTYPE", "lib", "lib_path", "ORBextractor", "ORBmatcher", "ComputeStereoMatches",
           "FeatSet", "make_featset", "STAGES", "BowDatabase", "DeviceKeyFrame", "BowFrames", "ORBVocabulary", "ComputeDistinctiveDescriptors", "Rectifier", "UndistortKeyPoints"]
