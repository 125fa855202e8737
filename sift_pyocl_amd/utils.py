"""Host helpers with the semantics of sift-src/utils.py:44-75 (calc_size, kernel_size, nextpower)."""
from math import ceil, log


def calc_size(shape, blocksize):
    """Round each dimension of `shape` up to a multiple of the (power-of-two) work-group size."""
    if hasattr(blocksize, "__len__"):
        return tuple((int(i) + int(j) - 1) & ~(int(j) - 1) for i, j in zip(shape, blocksize))
    return tuple((int(i) + int(blocksize) - 1) & ~(int(blocksize) - 1) for i in shape)


def kernel_size(sigma, odd=False, cutoff=4):
    """Number of taps of the Gaussian of width sigma: ceil(2*cutoff*sigma + 1), made odd on request."""
    size = int(ceil(2 * cutoff * sigma + 1))
    if odd and size % 2 == 0:
        size += 1
    return size


def nextpower(n):
    """Smallest power of two >= n."""
    return 1 << int(ceil(log(n, 2)))
