"""Console + file logger with the reference's interface (Utils/Log.py): Log(name, file).info(msg)."""
import logging
import os
import time


class Log:
    def __init__(self, module: str, filename: str, log_dir: str = "logs"):
        self.logger = logging.getLogger(f"diffmm_amd.{module}.{filename}")
        self.logger.setLevel(logging.INFO)
        self.logger.propagate = False
        if not self.logger.handlers:
            fmt = logging.Formatter("%(asctime)s %(message)s")
            sh = logging.StreamHandler()
            sh.setFormatter(fmt)
            self.logger.addHandler(sh)
            try:
                os.makedirs(log_dir, exist_ok=True)
                fh = logging.FileHandler(os.path.join(log_dir, f"{time.strftime('%Y%m%d-%H%M%S')}_{filename}.log"))
                fh.setFormatter(fmt)
                self.logger.addHandler(fh)
            except OSError:
                pass

    def info(self, msg) -> None:
        self.logger.info(msg)
