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


def matching_correction(matching):
    """Least-squares affine map sending the keypoints matching[:, 0] onto matching[:, 1].

    Model of sift-src/utils.py:156-189 (x' = a*x + b*y + c ; y' = d*x + e*y + f): the reference snapshot
    builds the (2N, 6) design matrix and the right-hand side and then stops -- the solve and the
    ``return`` are missing from the file -- so LinearAlign.align cannot run past it there.  The system is
    solved here in float64 with numpy.linalg.lstsq; returns the 6 parameters (a, b, c, d, e, f).
    """
    import numpy
    N = matching.shape[0]
    X = numpy.zeros((2 * N, 6))
    X[::2, 0] = matching.x[:, 0]
    X[::2, 1] = matching.y[:, 0]
    X[::2, 2] = 1
    X[1::2, 3] = matching.x[:, 0]
    X[1::2, 4] = matching.y[:, 0]
    X[1::2, 5] = 1
    y = numpy.zeros((2 * N,))
    y[::2] = matching.x[:, 1]
    y[1::2] = matching.y[:, 1]
    sol = numpy.linalg.lstsq(X, y, rcond=None)[0]
    return sol
