"""sloika/config.py:1-3 -- the reference takes its dtype from theano.config.floatX; here it is fixed."""
sloika_dtype = 'float32'
