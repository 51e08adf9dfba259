"""Drop-in for UPFlow/utils/loss.py `loss_functions`: the photometric tail and the census loss on
the fused HIP kernels."""
from ... import ops


class loss_functions:
    @classmethod
    def photo_loss_function(cls, diff, mask, q, charbonnier_or_abs_robust, if_use_occ, averge=True):
        """UPFlow/utils/loss.py:17-48."""
        return ops.photo_loss_function(diff, mask, q, charbonnier_or_abs_robust, if_use_occ, averge)

    @classmethod
    def census_loss_torch(cls, img1, img1_warp, mask, q, charbonnier_or_abs_robust, if_use_occ,
                          averge=True, max_distance=3):
        """UPFlow/utils/loss.py:51-91."""
        return ops.census_loss(img1, img1_warp, mask, q, charbonnier_or_abs_robust, if_use_occ, averge,
                               max_distance)
