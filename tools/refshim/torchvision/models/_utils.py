def _make_divisible(v, divisor, min_value=None):
    # standard MobileNet channel rounding rule
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v
