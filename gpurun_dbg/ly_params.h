// Parameter structs shared by kernels and the public C ABI.
#pragma once
#include "../include/lead_yolo_hip.h"
