import logging


def get_logger(name, log_file=None, log_level=logging.INFO):
    return logging.getLogger(name)
