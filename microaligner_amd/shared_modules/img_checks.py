"""Input validation with the reference's error behaviour (microaligner/shared_modules/img_checks.py:26-47):
ValueError for >2-D input, for a missing image and for mismatched shapes."""


def _shape(img):
    return tuple(img.shape)


def check_img_is_2d_grey(img, img_type: str):
    if len(_shape(img)) > 2:
        raise ValueError(
            f"Expected {img_type} input to be 2D grayscale image, "
            f"but received {img_type} image with shape {_shape(img)}")


def check_img_is_provided(img, img_type: str):
    if len(img) == 0:
        raise ValueError(f"No {img_type} image provided")


def check_img_dims_match(ref, mov):
    if _shape(ref) != _shape(mov):
        raise ValueError(
            "Input images have different dimensions "
            f"reference image shape: {_shape(ref)}, moving image shape: {_shape(mov)}")
