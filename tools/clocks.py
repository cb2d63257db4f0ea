"""Engine clock, memory clock and board power of a GPU as the kernel driver publishes them in sysfs, sampled by a host thread while a leg runs
(bench.py: the sustained launches; tools/bench_configs.py: the SuBSENSE block)."""
import os
import time


class ClockSampler:
    """Engine clock, memory clock and board power of this rank's GPU as the kernel driver publishes them in sysfs, read every few
    milliseconds by a host thread while the SUSTAINED launches run (the same step as the timed region, outside the contract's clock so
    that the reader cannot disturb it).  Boxes of the pool differ by 5-12 % on the same kernel; with the copy calibration this says
    whether a slow line is a slow clock.  Nothing here is required: whatever cannot be read is reported as such."""

    def __init__(self, local):
        import glob
        import threading
        self.samples, self.err, self.stop_flag, self.thread = [], None, False, None
        try:
            import torch
            pr = torch.cuda.get_device_properties(local)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            base = "/sys/bus/pci/devices/" + bdf
            hw = glob.glob(base + "/hwmon/hwmon*")
            self.files = {"sclk_hz": hw[0] + "/freq1_input" if hw else None, "mclk_hz": hw[0] + "/freq2_input" if hw else None,
                          "power_uw": next((hw[0] + "/" + n for n in ("power1_average", "power1_input") if os.path.exists(hw[0] + "/" + n)), None) if hw else None, "pp_sclk": base + "/pp_dpm_sclk", "pp_mclk": base + "/pp_dpm_mclk"}
            self.bdf = bdf
            if not os.path.isdir(base):
                self.err = "no sysfs node for %s" % bdf
        except Exception as ex:  # noqa: BLE001 - diagnostics only
            self.err = repr(ex)
        self._threading = threading

    @staticmethod
    def _read(path):
        try:
            return open(path).read()
        except Exception:  # noqa: BLE001
            return None

    @staticmethod
    def _starred_mhz(text):
        for line in (text or "").splitlines():
            if line.rstrip().endswith("*"):
                digits = "".join(ch for ch in line.split(":")[-1] if ch.isdigit())
                return int(digits) if digits else None
        return None

    def _one(self):
        f = self.files
        sclk = self._read(f["sclk_hz"]) if f["sclk_hz"] else None
        mclk = self._read(f["mclk_hz"]) if f["mclk_hz"] else None
        pw = self._read(f["power_uw"]) if f["power_uw"] else None
        rec = {"sclk": int(sclk) / 1e6 if sclk and sclk.strip().isdigit() else self._starred_mhz(self._read(f["pp_sclk"])),
               "mclk": int(mclk) / 1e6 if mclk and mclk.strip().isdigit() else self._starred_mhz(self._read(f["pp_mclk"])),
               "power": int(pw) / 1e6 if pw and pw.strip().isdigit() else None}
        self.samples.append(rec)

    def start(self):
        if self.err:
            return
        def run():
            while not self.stop_flag:
                self._one()
                time.sleep(0.004)
        self.thread = self._threading.Thread(target=run, daemon=True)
        self.thread.start()

    def stop(self):
        if self.thread:
            self.stop_flag = True
            self.thread.join(timeout=1.0)
        if self.err:
            return {"error": self.err}
        out = {"pci": self.bdf, "samples": len(self.samples), "source": "sysfs hwmon freq1_input / freq2_input / power1_average (or power1_input), else the starred level of pp_dpm_sclk / pp_dpm_mclk; read during the sustained launches"}
        for k, name in (("sclk", "engine_clock_MHz"), ("mclk", "memory_clock_MHz"), ("power", "board_power_W")):
            v = sorted(x[k] for x in self.samples if x[k] is not None)
            out[name] = {"min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)} if v else None
        return out
