#include "dbow2_stub.h"
