"""Independent float64 models of the T41's optional noise reduction / notch stages, written from the
algorithms' descriptions (Kim & Ruwisch 2002 spectral weighting; Romanin / Gerkmann-Hendriks MMSE with
speech-presence probability; WDSP's variable-leak LMS) with the reference's parameters and its
documented quirks (SURVEY-style list in oracle/t41_nr_oracle.c), NOT from the oracle's C text:
numpy FFTs, whole-array operations, no f32 rounding.  tests/test_noise_reduction.py checks the
oracle against them (catches indexing, pointer and state-flow mistakes in the restatement).
Blocks are 256 audio samples @24 kS/s."""
import numpy as np

N = 256
H = N // 2
SQRT_HANN = None  # set by the test from the oracle-independent formula sin(pi i / 255) (the table's definition)


def vad_range(FLoCut, FHiCut):
    if FLoCut <= 0 <= FHiCut:
        lf, uf = 0.0, float(max(-FLoCut, FHiCut))
    elif FLoCut > 0:
        lf, uf = float(FLoCut), float(FHiCut)
    else:
        lf, uf = float(-FHiCut), float(-FLoCut)
    binw = 24000.0 / N
    lo, hi = int(lf / binw), int(uf / binw)
    if lo == hi:
        hi += 1
    lo = min(max(lo, 1), H - 2)
    hi = min(max(hi, 1), H)
    return lo, hi


class Kim:
    """Kim1_NR: 256-point frames hopping by 128, Hann analysis window, power spectra averaged over 3
    frames (E), noise floor = minimum of the last 15 E (M), gain 1 - M / E clamped at 0, smoothed in
    time (alpha) and over neighbouring bins (beta), applied to bins i and 255 - i, overlap-add of the
    real part without synthesis window."""

    def __init__(self, FLoCut, FHiCut, alpha=0.95, beta=0.85, psi=0.0):
        self.lo, self.hi = vad_range(FLoCut, FHiCut)
        self.alpha, self.beta, self.psi = float(np.float32(alpha)), float(np.float32(beta)), float(np.float32(psi))
        self.X = np.zeros((H, 3))
        self.X[:, 1] = 0.5
        self.E = np.zeros((H, 15))
        self.E[:, 0] = 0.1
        self.Gts = np.zeros(H)          # NR_Gts[.][0] as left by the previous frame (bins outside the range keep this)
        self.Gts_prev = np.full(H, 0.1)  # NR_Gts[.][1]
        self.last_in = np.full(H, 0.1)
        self.last_out = np.zeros(H)
        self.xp = self.ep = 0
        idx = np.arange(N)
        self.win = 0.5 * (1.0 - np.cos(2.0 * np.pi * idx / (N - 1)))

    def block(self, x):
        x = np.asarray(x, dtype=np.float64)
        out = np.empty(N)
        lo, hi = self.lo, self.hi
        for k in range(2):
            new = x[k * H:(k + 1) * H]
            frame = np.concatenate([self.last_in, new]) * self.win
            self.last_in = new.copy()
            F = np.fft.fft(frame)
            self.X[:, self.xp] = np.abs(F[:H]) ** 2
            self.E[lo:hi, self.ep] = self.X[lo:hi].sum(axis=1) / 3.0
            M = self.E[lo:hi].min(axis=1)
            with np.errstate(divide="ignore", invalid="ignore"):
                T = self.X[lo:hi, self.xp] / M
                lam = np.where(T > self.psi, M, self.E[lo:hi, self.ep])
                G = 1.0 - lam / self.E[lo:hi, self.ep]
            G = np.where(G < 0.0, 0.0, G)
            self.Gts[lo:hi] = self.alpha * self.Gts_prev[lo:hi] + (1.0 - self.alpha) * G
            self.Gts_prev[lo:hi] = self.Gts[lo:hi]
            g = self.Gts
            Gs = np.empty(H)
            Gs[1:-1] = self.beta * g[:-2] + (1.0 - 2.0 * self.beta) * g[1:-1] + self.beta * g[2:]
            Gs[0] = (1.0 - self.beta) * g[0] + self.beta * g[1]
            Gs[-1] = self.beta * g[-2] + (1.0 - self.beta) * g[-1]
            F[:H] *= Gs
            F[N - 1 - np.arange(H)] *= Gs  # the partner the reference weights: bin 255 - i
            self.xp = (self.xp + 1) % 3
            self.ep = (self.ep + 1) % 15
            y = np.fft.ifft(F).real
            out[k * H:(k + 1) * H] = y[:H] + self.last_out
            self.last_out = y[H:].copy()
        return out


class Spectral:
    """SpectralNoiseReduction: sqrt-Hann analysis and synthesis windows, noise PSD tracked with a
    speech-presence probability, decision-directed a-priori SNR, MMSE-style gain, musical-noise
    smoothing whose width depends on the ratio of output to input power -- evaluated, as the reference
    does, after every single bin's gain update.  The first 20 half-blocks only initialise the noise
    estimate and leave the audio untouched."""

    def __init__(self, FLoCut, FHiCut, alpha=0.95):
        self.lo, self.hi = vad_range(FLoCut, FHiCut)
        self.alpha = float(np.float32(alpha))
        f = lambda v: float(np.float32(v))  # noqa: E731  (the reference's constants are f32 variables)
        self.tinc, self.tax, self.tap = f(0.00533333), f(0.0239), f(0.05062)
        self.psthr, self.pnsaf, self.psini = f(0.99), f(0.01), f(0.5)
        self.ax = float(np.exp(np.float32(-self.tinc / self.tax), dtype=np.float32))
        self.ap = float(np.exp(np.float32(-self.tinc / self.tap), dtype=np.float32))
        xih1 = 100.0
        self.xih1r = f(1.0 / (1.0 + xih1) - 1.0)
        self.pfac = f((1.0 / 0.5 - 1.0) * (1.0 + xih1))
        self.snr_prio_min = f(0.1)
        self.state = 1
        self.count = 0
        self.last_in = np.full(H, 0.1)
        self.last_out = np.zeros(H)
        self.G = np.zeros(H)
        self.Hk_old = np.full(H, 0.1)
        self.Nest = np.full(H, 0.01)
        self.pslp = np.zeros(H)
        self.xt = np.zeros(H)

    def block(self, x):
        x = np.asarray(x, dtype=np.float64).copy()
        lo, hi = self.lo, self.hi
        if self.state == 1:
            self.last_in[:] = 0.0
            self.G[:] = 1.0
            self.Hk_old[:] = 1.0
            self.Nest[:] = 0.0
            self.pslp[:] = 0.5
            self.state = 2
        for k in range(2):
            new = x[k * H:(k + 1) * H].copy()
            F = np.fft.fft(np.concatenate([self.last_in, new]) * SQRT_HANN)
            self.last_in = new
            X = np.abs(F[:H]) ** 2
            if self.state == 2:
                self.Nest += 0.05 * X
                self.xt = self.psini * self.Nest
                self.count += 1
                if self.count > 19:
                    self.count = 0
                    self.state = 3
            if self.state != 3:
                continue
            with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
                ph = 1.0 / (1.0 + self.pfac * np.exp(self.xih1r * X / self.xt))
            self.pslp = self.ap * self.pslp + (1.0 - self.ap) * ph
            ph = np.where(self.pslp > self.psthr, 1.0 - self.pnsaf, np.minimum(ph, 1.0))
            xtr = (1.0 - ph) * X + ph * self.xt
            self.xt = self.ax * self.xt + (1.0 - self.ax) * xtr
            post = np.maximum(np.minimum(X / self.xt, 1000.0), self.snr_prio_min)
            prio = np.maximum(self.alpha * self.Hk_old + (1.0 - self.alpha) * np.maximum(post - 1.0, 0.0), 0.0)
            G = self.G
            for i in range(lo, hi):
                v = prio[i] * post[i] / (1.0 + prio[i])
                G[i] = np.sqrt(0.7212 * v + v * v) / post[i]
                self.Hk_old[i] = post[i] * G[i] * G[i]
                pre = X[lo:hi].sum()
                pst = (G[lo:hi] ** 2 * X[lo:hi]).sum()
                ratio = pst / pre
                NN = 1 if ratio > float(np.float32(0.4)) else 1 + 2 * int(0.5 + 4 * (1.0 - ratio / float(np.float32(0.4))))
                h = NN // 2
                if NN > 1:
                    # centred moving average over the inner bins -- except the last NN - h of them, where the
                    # reference's "upper edge" pass (a backward average over NN bins) overwrites the centred
                    # value before the copy back into the gains
                    mid = np.array([G[j - h:j + h + 1].sum() / NN if j < hi - NN else G[j - NN + 1:j + 1].sum() / NN
                                    for j in range(lo + h, hi - h)])
                    G[lo + h:hi - h] = mid
            F[:H] *= G
            F[N - 1 - np.arange(H)] *= G
            y = np.fft.ifft(F).real * SQRT_HANN
            x[k * H:(k + 1) * H] = y[:H] + self.last_out
            self.last_out = y[H:].copy()
        return x


class Anr:
    """Xanr: 64-tap leaky normalised LMS predictor on a delay line (prediction distance 16 samples);
    y = the predictable part (noise-reduction output), input - y = the notch output.  The leak index
    is pinned at its minimum by the reference's if / else-if nesting, so the leak is constant."""

    def __init__(self):
        self.d = np.zeros(512)
        self.w = np.zeros(64)
        self.pos = 0
        f = lambda v: float(np.float32(v))  # noqa: E731
        self.two_mu = f(0.0001)
        lidx = 120.0
        self.ngamma0 = f(0.001)
        self.ngamma = float(np.float32(np.float32(np.float32(0.1) * np.float32(lidx * lidx)) * np.float32(lidx * lidx)) * np.float32(6.25e-10))
        self.first = True

    def block(self, x, notch):
        x = np.asarray(x, dtype=np.float64)
        out = np.empty(len(x))
        for i, s in enumerate(x):
            self.d[self.pos] = s
            win = self.d[(self.pos + 16 + np.arange(64)) & 511]
            y = float(self.w @ win)
            sigma = float(win @ win)
            inv = 1.0 / (sigma + 1e-10)
            err = s - y
            out[i] = err if notch else y
            c0 = 1.0 - self.two_mu * self.ngamma
            self.w = c0 * self.w + (self.two_mu * err * inv) * win
            self.pos = (self.pos - 1) & 511
        return out


def run(x, FLoCut, FHiCut, nrOptionSelect=0, ANR_notchOn=0, alpha=0.95, beta=0.85, psi=0.0):
    """Process.cpp:841-866 over a stream of 256-sample blocks"""
    x = np.asarray(x, dtype=np.float64)
    kim, spec, anr = Kim(FLoCut, FHiCut, alpha, beta, psi), Spectral(FLoCut, FHiCut, alpha), Anr()
    out = np.empty_like(x)
    for b in range(len(x) // N):
        blk = x[b * N:(b + 1) * N]
        if nrOptionSelect == 1:
            blk = 30.0 * kim.block(blk)
        elif nrOptionSelect == 2:
            blk = spec.block(blk)
        elif nrOptionSelect == 3:
            anr.block(blk, notch=False)  # its output is discarded by the call site
            blk = 1.5 * blk
        if ANR_notchOn:
            blk = anr.block(blk, notch=True)
        out[b * N:(b + 1) * N] = blk
    return out
