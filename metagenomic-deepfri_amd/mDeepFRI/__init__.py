"""mDeepFRI -- MI355X-native drop-in for the per-protein inference hot path of Metagenomic-DeepFRI.

Same import surface as the reference for this path (and nothing else):
    from mDeepFRI.contact_map_utils import align_contact_map, pairwise_sqeuclidean   (reference bio_utils.py:36)
    from mDeepFRI.predict import Predictor, seq2onehot                               (reference pipeline.py:39)
    from mDeepFRI.contact_map import CAlphaCoordinates, DistanceMap, ContactMap
    from mDeepFRI.bio_utils import calculate_contact_map, build_align_contact_map
plus the batched, multi-GPU counterparts the reference lacks (mDeepFRI.batch, mDeepFRI.sharding).
All compute runs in hand-written HIP kernels behind the C ABI of include/mdfri.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
