#!/opt/conda/bin/python3.9
"""The reference's OWN demo image (README.md:46-52) from the UNMODIFIED gpet_utils.construct_test_img, whose noise is
scikit-image's random_noise(..., seed=1) (gpet_utils.py:251).  The build container's second interpreter
(/opt/conda/bin/python3.9: numpy 1.26.4, scipy 1.7.1, scikit-image 0.18.3) imports that file as it is; the main interpreter
has no scikit-image.  Run in the build container only:

    /opt/conda/bin/python3.9 tests/golden/make_readme_image.py      # -> tests/golden/readme_image.npz (this script)
    python tests/golden/make_fixtures.py readme_trace                # -> the reference's trace of it (make_fixtures.py)

Stored: the image (float64, what comp_grad_img receives), the true edge, the gradient image of the unmodified
comp_grad_img (scipy.ndimage of THAT interpreter: a second pin of a1), and two more ltypes' images as hashes."""
import hashlib
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_gpet_utils", "/root/reference/gp_edge_tracing/gpet_utils.py")
ref_utils = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_utils)
import skimage  # noqa: E402

N = 500
img, edge = ref_utils.construct_test_img(size=(N, N), amplitude=200, curvature=4, noise_level=0.05, ltype='sinusoidal',
                                         intensity=0.3, gaps=True)
kernel = ref_utils.kernel_builder(size=(11, 5), unit=False)
grad = ref_utils.comp_grad_img(img, kernel)
out = dict(ref_img=img, ref_true_edge=edge, ref_grad_py39=grad, in_kernel=kernel,
           versions=np.array("skimage %s numpy %s" % (skimage.__version__, np.__version__)))
# other image types / sizes of the same generator: sha256 of the float64 bytes (the package's generator must reproduce them)
names, digests = [], []
for size, amp, curv, var, ltype, inten, gaps in [((96, 128), 40, 4, 0.02, 'multi-sinusoidal', 0.3, True),
                                                 ((128, 128), 60, 2, 0.05, 'co-sinusoidal', 0.4, False),
                                                 ((64, 64), 20, 4, 0.1, 'diag', 0.3, False),
                                                 ((64, 80), 20, 4, 0.01, 'straight', 0.3, True),
                                                 ((100, 100), 300, 3, 0.05, 'close multi-sinusoidal', 0.25, True)]:
    im, ed = ref_utils.construct_test_img(size=size, amplitude=amp, curvature=curv, noise_level=var, ltype=ltype, intensity=inten, gaps=gaps)
    names.append(repr((size, amp, curv, var, ltype, inten, gaps)))
    digests.append(hashlib.sha256(np.ascontiguousarray(im).tobytes()).hexdigest() + ":" + hashlib.sha256(np.ascontiguousarray(ed.astype(np.int64)).tobytes()).hexdigest())
out["other_args"] = np.array(names)
out["other_sha256"] = np.array(digests)
np.savez_compressed(os.path.join(HERE, "readme_image.npz"), **out)
print("readme_image.npz:", img.shape, img.dtype, "edge", edge.shape, "grad", grad.dtype, out["versions"])
