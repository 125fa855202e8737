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


def affine_least_squares(x, y, xp, yp):
    """Least-squares (a, b, c, d, e, f) of x' = a*x + b*y + c ; y' = d*x + e*y + f.

    Centred normal equations in float64: the two rows share one 3x3 system whose translation row decouples once
    the coordinates are centred, so the cost is a handful of dot products (O(N)) instead of an SVD of a (2N, 6)
    matrix; on pixel coordinates it agrees with numpy.linalg.lstsq to ~1e-12.
    """
    import numpy
    x = numpy.asarray(x, numpy.float64); y = numpy.asarray(y, numpy.float64)
    xp = numpy.asarray(xp, numpy.float64); yp = numpy.asarray(yp, numpy.float64)
    n = x.shape[0]
    mx, my = x.mean(), y.mean()
    xc = x - mx; yc = y - my

    def dot(u, v):
        # numpy's own pairwise summation, not BLAS ddot: a threaded BLAS wakes one worker per core for these 200 k-element
        # products and lets them spin afterwards -- on the GPU box (256 logical CPUs) the next 20-70 ms of the process,
        # its HIP submissions included, ran late (LinearAlign.align 103 ms with ddot, 55 ms without, round 4)
        return float((u * v).sum())

    A = numpy.array([[dot(xc, xc), dot(xc, yc)], [dot(xc, yc), dot(yc, yc)]])
    if n < 3 or abs(numpy.linalg.det(A)) <= 1e-12 * max(1.0, A[0, 0] * A[1, 1]):
        # degenerate geometry (collinear / too few points): fall back to the general solver
        X = numpy.zeros((2 * n, 6))
        X[::2, 0] = x; X[::2, 1] = y; X[::2, 2] = 1
        X[1::2, 3] = x; X[1::2, 4] = y; X[1::2, 5] = 1
        rhs = numpy.zeros((2 * n,)); rhs[::2] = xp; rhs[1::2] = yp
        return numpy.linalg.lstsq(X, rhs, rcond=None)[0]
    ab = numpy.linalg.solve(A, numpy.array([dot(xc, xp), dot(yc, xp)]))
    de = numpy.linalg.solve(A, numpy.array([dot(xc, yp), dot(yc, yp)]))
    c = xp.mean() - ab[0] * mx - ab[1] * my
    f = yp.mean() - de[0] * mx - de[1] * my
    return numpy.array([ab[0], ab[1], c, de[0], de[1], f])


def matching_correction(matching):
    """Least-squares affine map sending the keypoints matching[:, 0] onto matching[:, 1].

    Model of sift-src/utils.py:156-189 (x' = a*x + b*y + c ; y' = d*x + e*y + f): the reference snapshot
    builds the (2N, 6) design matrix and the right-hand side and then stops -- the solve and the
    ``return`` are missing from the file -- so LinearAlign.align cannot run past it there.  The system is
    solved here in float64 (affine_least_squares); returns the 6 parameters (a, b, c, d, e, f).
    """
    return affine_least_squares(matching.x[:, 0], matching.y[:, 0], matching.x[:, 1], matching.y[:, 1])
