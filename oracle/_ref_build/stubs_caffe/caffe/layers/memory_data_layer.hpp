// see ../caffe.hpp (stand-in, test infrastructure)
#pragma once
#include "caffe/caffe.hpp"
