#include "cv_stub.h"
