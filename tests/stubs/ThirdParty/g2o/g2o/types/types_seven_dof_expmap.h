#include "g2o_stub.h"
