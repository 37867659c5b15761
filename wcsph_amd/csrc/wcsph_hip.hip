// wcsph_hip.hip -- C ABI (include/wcsph.h) over the gfx950 kernels in grid.h / dfsph_kernels.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see __graft_entry__.build()).
#include "../../include/wcsph.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "device.h"
#include "grid.h"
#include "dfsph_kernels.h"
#include "solver_extra.h"
#include "surface.h"
#include "sampler.h"
#include "mcgrid.h"
#include "canvas.h"
#include "bucket.h"

#include "api_core.h"
#include "api_setup.h"
#include "api_grid.h"
#include "api_fields.h"
#include "api_dfsph.h"
#include "api_solvers.h"
#include "api_post.h"
#include "api_comm.h"
#include "api_slab.h"
#include "api_migrate.h"
#include "api_measure.h"
