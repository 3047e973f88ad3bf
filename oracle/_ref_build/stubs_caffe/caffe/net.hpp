// see caffe.hpp in this directory (stand-in, test infrastructure)
#pragma once
#include "caffe/caffe.hpp"
