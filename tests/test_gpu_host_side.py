"""The host-side pins (BVH4 builder, scene loader, oracle against the published images, image
decoders and writers, the plain-float acos/atan2/sin/cos, the exported C ABI, the 2-rank CPU
rehearsal of the multi-GPU splits, bench.py's launcher, the float32 check of the far-face safe distance) once more under the `gpu` marker: they need
no GPU, but the GPU-marked run is the one that is repeated on the MI355X machine, and a parity claim
made there rests on these as much as on the device tests."""
import pytest

from test_bvh_build import *            # noqa: F401,F403
from test_materials_scene import test_oracle_reproduces_published_materials_render, test_scene_indices  # noqa: F401
from test_normal_map_scene import test_oracle_reproduces_published_images  # noqa: F401
from test_oracle_golden import *        # noqa: F401,F403
from test_scene_loader import *         # noqa: F401,F403
from test_image_decode import *         # noqa: F401,F403
from test_outputs import *              # noqa: F401,F403
from test_det_math import *             # noqa: F401,F403
from test_capi_symbols import *         # noqa: F401,F403
from test_multi_gpu_cpu import *        # noqa: F401,F403
from test_bench_launcher import *       # noqa: F401,F403
from test_far_faces import *            # noqa: F401,F403

pytestmark = pytest.mark.gpu
