import sys
sys.path.insert(0, "/root/repo")
from tests.common import bench_setup, golden_bench_outputs, pair_digest
from gan_segmentation_amd.image_generator import ImageGenerator
gcfg, gp, dcfg, dp, z, noise = bench_setup("ffhq", 4)
gen = ImageGenerator.from_params(gcfg, gp, dcfg, dp, gpu_ids=[0], batch_size=4)
img, mask = gen.generate_batch(z, noise)
img, mask = img.cpu().numpy(), mask.cpu().numpy()
print("digest ok:", pair_digest(img[0], mask[0]) == golden_bench_outputs()["ffhq_b4"]["samples"][0])
