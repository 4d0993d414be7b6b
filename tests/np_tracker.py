"""Independent Python restatement of the tracker's LIST LOGIC -- what lives on the host in the product (csrc/host_logic.hpp, the host half of
csrc/stabilizer.hip::track) and in oracle/stabilizer.cpp: FeatureDetector::configure / detect / propagate / reset with its suppression grid,
detection regions and adaptive FAST thresholds (Vision/FeatureDetector.cpp:48-215), SpatialMap::distribution_quality (Data/SpatialMap.tpp:
589-625), VirtualGrid::key_of / test_point (Math/VirtualGrid.cpp:150-196), fast_filter / fast_erase (Functions/Container.tpp:30-121) and the
control flow of FrameTracker::track (Vision/FrameTracker.cpp:108-196).  Written from the reference's sources, NOT from the oracle or the
product (round-2 VERDICT, weak #1 (ii): those two are near-twins by one hand).

The pixel / numeric kernels are taken as given building blocks -- each has its own parity tests: the INTER_AREA downscale, FAST-9/16, the
pyramidal LK flow (tests/np_pyrlk.py is its independent restatement), the global motion estimate and the mesh least squares
(tests/test_mesh_lstsq.py) -- and are passed in as callables.  tests/test_np_tracker.py runs this model beside oracle's whole stabilizer and
compares the feature lists (position, response, age), counts, distribution quality and stability frame by frame.  Test infrastructure only."""
import numpy as np

F = np.float32
FAST_MIN_THRESHOLD, FAST_MAX_THRESHOLD, FAST_THRESHOLD_STEP, FAST_FEATURE_TOLERANCE = 10, 250, 5, 150       # FeatureDetector.cpp:28-31
HOMOGRAPHY_DISTRIBUTION_THRESHOLD = F(0.6)                                                                  # FrameTracker.cpp:37
SIZE_T = 1 << 64


def cv_round(x):
    return int(np.rint(F(x)))                 # saturate_cast<int>(float): round half to even


def step(current, target, amount):            # Functions/Math.tpp:133-142
    return max(current - amount, target) if current > target else min(current + amount, target)


class Grid:
    """VirtualGrid: resolution + float alignment rectangle (Math/VirtualGrid.cpp:85-91,150-196)."""

    def __init__(self, cols, rows, x, y, w, h):
        self.cols, self.rows = cols, rows
        self.x, self.y, self.w, self.h = F(x), F(y), F(w), F(h)
        self.kw, self.kh = F(self.w / F(cols)), F(self.h / F(rows))

    def test_point(self, px, py):             # cv::Rect2f::contains
        return self.x <= px < F(self.x + self.w) and self.y <= py < F(self.y + self.h)

    def key_of(self, px, py):                 # static_cast<size_t> of a non-negative float quotient: truncation
        return int(F(F(px - self.x) / self.kw)), int(F(F(py - self.y) / self.kh))


class Feature:
    __slots__ = ("x", "y", "response", "age")

    def __init__(self, x, y, response, age=0):
        self.x, self.y, self.response, self.age = F(x), F(y), F(response), int(age)

    def copy(self):
        return Feature(self.x, self.y, self.response, self.age)


class Detector:
    def __init__(self, s, fast):
        """s: settings with the reference's field names; fast(image, (x, y, w, h), threshold) -> [(x, y, score)] ROI-local, row-major."""
        self.fast = fast
        self.configure(s)

    def configure(self, s):                   # FeatureDetector.cpp:48-83
        dw, dh = s.detection_width, s.detection_height
        gc, gr = cv_round(F(dw) * F(s.max_feature_density)), cv_round(F(dh) * F(s.max_feature_density))
        self.grid = Grid(gc, gr, 0, 0, dw, dh)
        self.cells = {}                       # suppression grid: key -> index into self.features (SpatialMap of size_t)
        self.regions_grid = Grid(s.detection_regions_x, s.detection_regions_y, 0, 0, dw, dh)
        rw, rh = self.regions_grid.kw, self.regions_grid.kh
        self.regions = []                     # construct_detection_regions :87-110, row by row
        for r in range(s.detection_regions_y):
            for c in range(s.detection_regions_x):
                self.regions.append({"bounds": (F(F(c) * rw), F(F(r) * rh), rw, rh), "threshold": FAST_MIN_THRESHOLD, "load": 0})
        max_features = gc * gr
        max_region_features = F(F(max_features) / F(len(self.regions)))
        density_ratio = F(F(s.min_feature_density) / F(s.max_feature_density))
        self.minimum_load = int(F(max_region_features * density_ratio))
        self.target = int(F(F(s.accumulation_rate) * max_region_features))
        self.force = bool(s.force_detection)
        self.features = []

    def detect(self, image):                  # :114-172
        for region in self.regions:
            if self.force or region["load"] <= self.minimum_load:
                bx, by, bw, bh = region["bounds"]
                roi = (cv_round(bx), cv_round(by), cv_round(bw), cv_round(bh))          # cv::Mat::operator()(Rect) of a Rect2f
                found = self.fast(image, roi, region["threshold"])
                for (x, y, score) in found:
                    f = Feature(F(F(x) + bx), F(F(y) + by), F(score), 0)
                    key = self.grid.key_of(f.x, f.y)
                    if key not in self.cells:
                        self.cells[key] = len(self.features)
                        self.features.append(f)
                    else:
                        best = self.features[self.cells[key]]
                        if f.response > best.response and best.age <= 0:
                            self.features[self.cells[key]] = f
                n = len(found)
                if n > self.target + FAST_FEATURE_TOLERANCE:
                    region["threshold"] = step(region["threshold"], FAST_MAX_THRESHOLD, FAST_THRESHOLD_STEP)
                elif n < (self.target - FAST_FEATURE_TOLERANCE) % SIZE_T:               # size_t arithmetic: wraps for small targets
                    region["threshold"] = step(region["threshold"], FAST_MIN_THRESHOLD, FAST_THRESHOLD_STEP)
            region["load"] = 0
        out, self.features = self.features, []
        quality = self.distribution_quality()
        self.cells = {}
        return out, quality

    def distribution_quality(self):           # SpatialMap.tpp:589-625
        n = len(self.cells)
        if n == 0:
            return F(1.0)
        cols, rows = self.grid.cols, self.grid.rows
        if cols <= 4 or rows <= 4:
            return F(F(n) / F(cols * rows))
        sectors = Grid(4, 4, 0, 0, cols, rows)
        buckets = [0] * 16
        ideal = int(F(F(n) / F(16)))
        excess = F(0.0)
        for (kx, ky) in self.cells:
            if sectors.test_point(F(kx), F(ky)):
                sx, sy = sectors.key_of(F(kx), F(ky))
                buckets[sy * 4 + sx] += 1
                if buckets[sy * 4 + sx] > ideal:
                    excess = F(excess + F(1.0))
        return F(F(1.0) - F(excess / F(n - ideal)))

    def propagate(self, features):            # :176-199
        for f in features:
            if not self.grid.test_point(f.x, f.y):
                continue
            key = self.grid.key_of(f.x, f.y)
            if key not in self.cells:
                self.cells[key] = len(self.features)
                rx, ry = self.regions_grid.key_of(f.x, f.y)
                self.regions[ry * self.regions_grid.cols + rx]["load"] += 1
                self.features.append(f.copy())
            else:
                best = self.features[self.cells[key]]
                if f.response > best.response and f.age >= best.age:
                    self.features[self.cells[key]] = f.copy()

    def reset(self):                          # :203-208 (m_Features is NOT cleared)
        self.cells = {}
        for region in self.regions:
            region["load"] = 0


def fast_erase(data, k):                      # Container.tpp:30-39
    data[k], data[-1] = data[-1], data[k]
    data.pop()


class Tracker:
    """FrameTracker::track (FrameTracker.cpp:108-196) over injected kernels:
       downscale(frame) -> tracking image; flow(prev, cur, points [n, 2]) -> (matched [n, 2], status [n]);
       estimate(tracked [m, 2], matched [m, 2], homography: bool) -> inlier flags [m] (and whatever the caller records about the model)."""

    def __init__(self, s, downscale, fast, flow, estimate):
        self.s = s
        self.downscale, self.flow, self.estimate = downscale, flow, estimate
        self.detector = Detector(s, fast)
        self.prev = self.cur = None
        self.initialized = False
        self.features = []
        self.stability = F(0.0)
        self.last = {}

    def restart(self):                        # :97-104
        self.stability = F(0.0)
        self.features = []
        self.detector.reset()
        self.initialized = False

    def track(self, frame):
        s = self.s
        self.stability = F(0.0)
        self.last = {"detected": 0, "matched": 0, "distribution": F(0.0), "estimated": False}
        self.prev, self.cur = self.cur, self.downscale(frame)
        if not self.initialized or self.prev is None or self.prev.shape != self.cur.shape:
            self.initialized = True
            return False
        self.features, distribution = self.detector.detect(self.cur)
        self.last["detected"], self.last["distribution"] = len(self.features), distribution
        if len(self.features) < s.min_motion_samples or distribution < F(s.uniformity_threshold):
            self.features = []
            return False
        tracked = [[f.x, f.y] for f in self.features]
        matched, status = self.flow(self.prev, self.cur, np.array(tracked, F).reshape(-1, 2))
        matched = [list(m) for m in matched]
        # fast_filter(features, tracked, matched, status): back to front, swap with the last, pop (Container.tpp:97-121)
        for k in range(len(status) - 1, -1, -1):
            if not status[k]:
                fast_erase(self.features, k); fast_erase(tracked, k); fast_erase(matched, k)
        self.last["matched"] = len(matched)
        if len(matched) < s.min_motion_samples:
            self.features = []
            return False
        tracked_a, matched_a = np.array(tracked, F).reshape(-1, 2), np.array(matched, F).reshape(-1, 2)
        inliers = self.estimate(tracked_a, matched_a, bool(distribution > HOMOGRAPHY_DISTRIBUTION_THRESHOLD))
        self.last["estimated"] = True
        self.last["pairs"] = (tracked_a, matched_a)
        self.stability = F(F(int(np.count_nonzero(np.asarray(inliers) == 1))) / F(len(inliers)))      # ratio_of<uint8_t>(status, 1)
        for i in range(len(inliers) - 1, -1, -1):                                                     # :183-192
            if inliers[i]:
                self.features[i].age += 1
                self.features[i].x, self.features[i].y = F(matched_a[i, 0]), F(matched_a[i, 1])
            else:
                fast_erase(self.features, i)
        self.detector.propagate(self.features)
        return True
