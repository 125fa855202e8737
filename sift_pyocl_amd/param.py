"""SIFT parameters -- same names, values and mutate-to-configure semantics as the reference's
``sift_pyocl.param.par`` (sift-src/param.py:43-79).  Values are read at call time by SiftPlan /
MatchPlan, so ``par["PeakThresh"] = ...`` takes effect on the next ``keypoints()`` call.
"""


class Enum(dict):
    """dict whose keys are also attributes (sift-src/param.py:43-50)."""

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)


par = Enum(OctaveMax=100000,          # never read by the reference either
           DoubleImSize=0,
           order=3,
           InitSigma=1.6,
           BorderDist=5,
           Scales=3,
           PeakThresh=255.0 * 0.04 / 3.0,
           EdgeThresh=0.06,
           EdgeThresh1=0.08,
           OriBins=36,
           OriSigma=1.5,
           OriHistThresh=0.8,
           MaxIndexVal=0.2,
           MagFactor=3,
           IndexSigma=1.0,
           IgnoreGradSign=0,
           MatchRatio=0.73,
           MatchXradius=1000000.0,
           MatchYradius=1000000.0,
           noncorrectlylocalized=0)
